"""Oracle (test infrastructure, NOT product code) for ``phase_correlation(backend="skimage")`` (signal/tracking.py:262-272).

PARITY UNPINNED: the reference delegates to ``skimage.registration.phase_cross_correlation(img_z, tpl_pad,
upsample_factor=10 or 1)``; scikit-image is absent from this image and from the wheelhouse, ``pyproject.toml`` lists it
unpinned, and the reference holds no vectors for this back-end.  This file restates the PUBLISHED algorithm of that function
(scikit-image >= 0.19 defaults: space="real", normalization="phase"; Guizar-Sicairos, Thurman & Fienup, Opt. Lett. 33, 156
(2008): coarse peak of the phase-normalised cross-correlation, then a matrix-multiply DFT of a 1.5-pixel neighbourhood
up-sampled `upsample_factor` times) in float64 NumPy.  Self-checks in tests/test_phase_skimage_oracle.py: integer shifts
exact, Fourier-shifted inputs recovered to the 0.1-px grid, agreement of the coarse stage with the internal back-end's arg-max.
"""
from __future__ import annotations

import numpy as np

from . import signal_np as S


def upsampled_dft(data, upsampled_region_size, upsample_factor, axis_offsets):
    """Matrix-multiply DFT of `data` on an up-sampled grid of `upsampled_region_size` points per axis starting at
    `axis_offsets` (in up-sampled pixels): one (region x n) kernel per axis, applied last axis first."""
    size = [int(upsampled_region_size)] * data.ndim
    for n_items, ups_size, ax_offset in list(zip(data.shape, size, axis_offsets))[::-1]:
        kernel = (np.arange(ups_size) - ax_offset)[:, None] * np.fft.fftfreq(n_items, upsample_factor)
        kernel = np.exp(-2j * np.pi * kernel)
        data = np.tensordot(kernel, data, axes=(1, -1))
    return data


def cross_power(reference_image, moving_image):
    """Phase-normalised cross-power spectrum F_ref conj(F_mov) / max(|.|, 100 eps) (normalization="phase")."""
    src = np.fft.fft2(np.asarray(reference_image))
    tgt = np.fft.fft2(np.asarray(moving_image))
    prod = src * tgt.conj()
    eps = np.finfo(prod.real.dtype).eps
    prod = prod / np.maximum(np.abs(prod), 100 * eps)
    return prod


def shift_from_cross_power(prod, upsample_factor=1):
    """(dy, dx) that registers the moving image with the reference one, from their normalised cross-power spectrum."""
    shape = prod.shape
    cc = np.fft.ifft2(prod)
    maxima = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)
    midpoint = np.array([np.fix(n / 2) for n in shape])
    ftype = prod.real.dtype
    shift = np.stack(maxima).astype(ftype, copy=False)
    shift[shift > midpoint] -= np.array(shape)[shift > midpoint]
    if upsample_factor > 1:
        uf = np.array(upsample_factor, dtype=ftype)
        shift = np.round(shift * uf) / uf
        region = np.ceil(uf * 1.5)
        dftshift = np.fix(region / 2.0)
        offset = dftshift - shift * uf
        cc = upsampled_dft(prod.conj(), region, uf, offset).conj()
        maxima = np.unravel_index(np.argmax(np.abs(cc)), cc.shape)
        maxima = np.stack(maxima).astype(ftype, copy=False) - dftshift
        shift = shift + maxima / uf
    for dim in range(prod.ndim):      # a one-pixel axis carries no shift
        if shape[dim] == 1:
            shift[dim] = 0
    return shift


def phase_cross_correlation(reference_image, moving_image, *, upsample_factor=1):
    """The shift part of skimage.registration.phase_cross_correlation (error and phase difference are not used by barc4dip)."""
    return shift_from_cross_power(cross_power(reference_image, moving_image), upsample_factor)


def phase_correlation_skimage(template, image, *, slices_yx=None, subpixel=True, eps=1e-9):
    """signal/tracking.py:241-272 with backend="skimage": z-scores, zero-embedded template, up-sampling factor 10 (or 1);
    returns (dy, dx, nan, nan) like the reference."""
    tpl = S.as_float2d(template, "template")
    img = S.as_float2d(image, "image")
    H, W = img.shape
    if slices_yx is None:
        slices_yx = S.roi_slices((H, W), tpl.shape, center_yx=None, clip=False)
    img_z = S.zscore2d(img, eps)
    tpl_pad = S.embed_roi(S.zscore2d(tpl, eps), out_shape=(H, W), slices_yx=slices_yx, fill_value=0.0, dtype=np.float32)
    shift = phase_cross_correlation(img_z, tpl_pad, upsample_factor=10 if subpixel else 1)
    return float(shift[0]), float(shift[1]), float("nan"), float("nan")
