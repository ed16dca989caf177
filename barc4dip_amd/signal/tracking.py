"""Translation tracking on the GPU -- drop-in for ``barc4dip.signal.tracking``.

``track_translation`` keeps the reference's registry/dispatcher (tracking.py:12-78): methods are
looked up by name in ``_TRACKERS`` and receive ``backend=``.  ``phase_correlation`` with
``backend="internal"`` (tracking.py:191-297) runs entirely on the device: z-scoring, zero
embedding, both forward transforms, whitened cross-power spectrum, inverse transform, |.|,
first-occurrence arg-max, peak, exact median SNR and the 3x3 Taylor step (including the
reference's swapped corrections, tracking.py:372-373).  ``phase_correlation_batch`` exposes
the batched form used for stacks: every distinct image and template is transformed once.

``template_matching`` (tracking.py:81-188) wraps cv2.matchTemplate(TM_CCOEFF_NORMED) /
skimage.feature.match_template in the reference.  Here both back-end names run the same zero-mean
normalised cross-correlation on the device (``b4d_template_match``): "opencv" z-scores the image as
the reference does before calling cv2, "skimage" passes the raw float32 image.  Parity with the two
libraries is unpinned (they are absent from the build image); the oracle follows their published
definition (oracle/ncc_np.py).

``phase_correlation(backend="skimage")`` (tracking.py:262-272: skimage.registration.phase_cross_correlation with a ten-fold
up-sampled peak search) is built from that function's published algorithm instead of raising ``ImportError``: device
transforms, host refinement (``_phase_correlation_upsampled``; oracle/phase_skimage_np.py, parity unpinned).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Literal

import numpy as np

from .. import _device as D
from .. import _ffi
from ..geometry.roi import roi_slices

_Tracker = Callable[..., tuple]
_TRACKERS: dict[str, _Tracker] = {}


def _register(method: str):
    key = method.strip().lower()

    def deco(fn):
        _TRACKERS[key] = fn
        return fn

    return deco


def track_translation(template, image, *, slices_yx=None, method: str = "phase",
                      backend: Literal["internal", "skimage"] = "internal", subpixel: bool = True,
                      eps: float = 1e-9):
    """Dispatcher for translation tracking methods -> (dy, dx, peak_value, snr)."""
    fn = _TRACKERS.get(method.strip().lower())
    if fn is None:
        raise ValueError(f"Unsupported tracking method: {method!r}. Supported: {', '.join(sorted(_TRACKERS))}")
    return fn(template, image, slices_yx=slices_yx, backend=backend, subpixel=subpixel, eps=eps)


def _as_float2d(a, *, name: str):
    if not D.is_tensor(a):
        a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError(f"{name} must be a 2D array.")
    return a


def _check_pairs(images, tpl_src, tpl_frame, tpl_roi, pair_img, pair_tpl):
    im, _, _ = D.to_device_f32(images, ndim=(3,))
    if tpl_src is images:
        ts = im
    else:
        ts, _, _ = D.to_device_f32(tpl_src, ndim=(3,))
    if tuple(ts.shape[1:]) != tuple(im.shape[1:]):
        raise ValueError("tpl_src frames must have the image shape.")
    tf = np.ascontiguousarray(tpl_frame, dtype=np.int32).ravel()
    tr = np.ascontiguousarray(tpl_roi, dtype=np.int32).reshape(-1, 4)
    pi = np.ascontiguousarray(pair_img, dtype=np.int32).ravel()
    pt = np.ascontiguousarray(pair_tpl, dtype=np.int32).ravel()
    if tf.size != tr.shape[0] or pi.size != pt.size:
        raise ValueError("index arrays have inconsistent lengths.")
    return im, ts, tf, tr, pi, pt


def _canvas(n: int) -> int:
    p = 64
    while p < n:
        p *= 2
    if p > 4096:
        raise NotImplementedError(f"frame side {n} exceeds the largest native transform (4096).")
    return p


def _pad_frames(t, ny: int, nx: int):
    torch = _ffi.require_gpu()
    out = torch.zeros((t.shape[0], ny, nx), dtype=torch.float32, device=t.device)
    out[:, :t.shape[1], :t.shape[2]] = t
    return out


def template_matching_batch(images, tpl_src, tpl_frame, tpl_roi, pair_img, pair_tpl, *, backend: str = "opencv",
                            subpixel: bool = True, eps: float = 1e-9, return_peak_ij: bool = False):
    """Batched NCC template matching; operands as in ``phase_correlation_batch``.  Returns (npairs, 4) float64 rows
    (dy, dx, peak, snr) [, (npairs, 2) int32 arg-max indices in the (ny-h+1, nx-w+1) match map]."""
    torch = _ffi.require_gpu()
    if backend not in ("opencv", "skimage"):
        raise ValueError("backend must be 'opencv' or 'skimage'.")
    im, ts, tf, tr, pi, pt = _check_pairs(images, tpl_src, tpl_frame, tpl_roi, pair_img, pair_tpl)
    nimg, H, W = (int(v) for v in im.shape)
    # the "valid" correlation is linear: frames of any size ride a zero-padded power-of-two canvas exactly
    ny, nx = _canvas(H), _canvas(W)
    if (ny, nx) != (H, W):
        same = ts is im
        im = _pad_frames(im, ny, nx)
        ts = im if same else _pad_frames(ts, ny, nx)
    npairs = int(pi.size)
    pl = _ffi.get_plan(ny, nx)
    out = torch.empty((npairs, 4), dtype=torch.float64, device=im.device)
    pij = torch.empty((npairs, 2), dtype=torch.int32, device=im.device)
    as_p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _ffi.check(_ffi.lib().b4d_template_match(
        pl.handle, D.ptr(im), int(nimg), D.ptr(ts), int(ts.shape[0]), as_p(tf), as_p(tr), int(tf.size),
        as_p(pi), as_p(pt), npairs, H, W, int(backend == "opencv"), int(bool(subpixel)), float(eps), D.ptr(out), D.ptr(pij),
        _ffi.stream_ptr()))
    res = out.cpu().numpy()
    return (res, pij.cpu().numpy()) if return_peak_ij else res


@_register("template")
def template_matching(template, image, *, slices_yx=None, backend: Literal["opencv", "skimage"] = "opencv",
                      subpixel: bool = True, eps: float = 1e-9):
    """Translation (dy, dx) by NCC template matching (reference: tracking.py:81-188) -> (dy, dx, peak_value, snr)."""
    torch = _ffi.require_gpu()
    tpl = _as_float2d(template, name="template")
    img = _as_float2d(image, name="image")
    H, W = img.shape
    h, w = tpl.shape
    if h > H or w > W:
        raise ValueError(f"template shape {(h, w)} must fit inside image shape {(H, W)}")
    if slices_yx is None:
        slices_yx = roi_slices((H, W), (h, w), center_yx=None, clip=False)
    sy, sx = slices_yx
    if backend not in ("opencv", "skimage"):
        raise ValueError("backend must be 'opencv' or 'skimage'.")
    # the kernels take the template as a ROI of a frame: put it at its reference position (or, if that position is not
    # inside the frame, at the origin -- only the ROI statistics and the reference centre enter the result)
    y0, x0 = int(sy.start), int(sx.start)
    inside = 0 <= y0 and y0 + h <= H and 0 <= x0 and x0 + w <= W and (sy.stop - sy.start, sx.stop - sx.start) == (h, w)
    py, px = (y0, x0) if inside else (0, 0)
    timg, _, _ = D.to_device_f32(img[None], ndim=(3,))
    ttpl, _, _ = D.to_device_f32(tpl, ndim=(2,))
    canvas = torch.zeros((1, H, W), dtype=torch.float32, device=timg.device)
    canvas[0, py:py + h, px:px + w] = ttpl
    r = template_matching_batch(timg, canvas, [0], [[py, py + h, px, px + w]], [0], [0], backend=backend, subpixel=subpixel,
                                eps=eps)[0]
    # centre-to-centre shift against the reference centre of slices_yx (tracking.py:142-143, 182-186)
    yc = (sy.start + sy.stop - 1) / 2.0
    xc = (sx.start + sx.stop - 1) / 2.0
    dy = float(r[0]) + py + (h - 1) / 2.0 - yc
    dx = float(r[1]) + px + (w - 1) / 2.0 - xc
    return dy, dx, float(r[2]), float(r[3])


def phase_correlation_batch(images, tpl_src, tpl_frame, tpl_roi, pair_img, pair_tpl, *, subpixel: bool = True,
                            eps: float = 1e-9, return_peak_ij: bool = False):
    """Batched phase correlation.

    images (nimg, ny, nx), tpl_src (nsrc, ny, nx): NumPy arrays or ROCm tensors (float32 used).
    tpl_frame (ntpl,), tpl_roi (ntpl, 4) = (y0, y1, x0, x1): template k is that ROI of tpl_src[tpl_frame[k]].
    pair_img, pair_tpl (npairs,): pair i correlates images[pair_img[i]] with template pair_tpl[i].
    Returns (npairs, 4) float64 rows (dy, dx, peak, snr) [, (npairs, 2) int32 arg-max indices]."""
    torch = _ffi.require_gpu()
    im, _, _ = D.to_device_f32(images, ndim=(3,))
    if tpl_src is images:
        ts = im
    else:
        ts, _, _ = D.to_device_f32(tpl_src, ndim=(3,))
    if tuple(ts.shape[1:]) != tuple(im.shape[1:]):
        raise ValueError("tpl_src frames must have the image shape.")
    nimg, ny, nx = im.shape
    tf = np.ascontiguousarray(tpl_frame, dtype=np.int32).ravel()
    tr = np.ascontiguousarray(tpl_roi, dtype=np.int32).reshape(-1, 4)
    pi = np.ascontiguousarray(pair_img, dtype=np.int32).ravel()
    pt = np.ascontiguousarray(pair_tpl, dtype=np.int32).ravel()
    if tf.size != tr.shape[0] or pi.size != pt.size:
        raise ValueError("index arrays have inconsistent lengths.")
    npairs = int(pi.size)
    pl = _ffi.get_plan(ny, nx)
    out = torch.empty((npairs, 4), dtype=torch.float64, device=im.device)
    pij = torch.empty((npairs, 2), dtype=torch.int32, device=im.device)
    as_p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _ffi.check(_ffi.lib().b4d_phase_correlation(
        pl.handle, D.ptr(im), int(nimg), D.ptr(ts), int(ts.shape[0]), as_p(tf), as_p(tr), int(tf.size),
        as_p(pi), as_p(pt), npairs, int(bool(subpixel)), float(eps), D.ptr(out), D.ptr(pij), _ffi.stream_ptr()))
    res = out.cpu().numpy()
    return (res, pij.cpu().numpy()) if return_peak_ij else res


def _upsampled_dft(data: np.ndarray, region: int, upsample_factor: float, offsets) -> np.ndarray:
    """Matrix-multiply DFT of `data` on `region` up-sampled points per axis from `offsets` (Guizar-Sicairos et al. 2008)."""
    for n_items, off in list(zip(data.shape, offsets))[::-1]:
        kernel = np.exp(-2j * np.pi * (np.arange(region) - off)[:, None] * np.fft.fftfreq(n_items, upsample_factor))
        data = np.tensordot(kernel, data, axes=(1, -1))
    return data


def _phase_correlation_upsampled(tpl, img, slices_yx, subpixel: bool, eps: float):
    """``backend="skimage"`` (tracking.py:262-272): the reference hands z-scored image and zero-embedded z-scored template to
    ``skimage.registration.phase_cross_correlation(..., upsample_factor=10 if subpixel else 1)`` and returns its shift with
    NaN for peak and snr.  Built here from that function's published algorithm (phase-normalised cross-power spectrum, coarse
    peak, matrix-multiply DFT of a 1.5-pixel neighbourhood on a 0.1-px grid) so that the back-end works without scikit-image:
    the two N x N transforms run on the device (``b4d_fft2d``); the cross-power spectrum, its inverse transform for the coarse
    peak and the 15 x 15 up-sampled neighbourhood are host NumPy -- a per-call convenience path, not a throughput path.
    Parity with scikit-image itself is unpinned (absent from the build image): the oracle is oracle/phase_skimage_np.py."""
    from .fft import fft2d_stack

    sy, sx = slices_yx
    if D.is_tensor(img):
        img = D.to_host(img)
    if D.is_tensor(tpl):
        tpl = D.to_host(tpl)
    img = img if np.issubdtype(img.dtype, np.floating) else img.astype(np.float32)
    tpl = tpl if np.issubdtype(tpl.dtype, np.floating) else tpl.astype(np.float32)
    H, W = img.shape
    img_z = (img - float(np.nanmean(img))) / (float(np.nanstd(img)) + eps)          # _zscore2d (tracking.py:308-311)
    tpl_pad = np.zeros((H, W), dtype=np.float32)                                    # embed_roi(fill 0, float32) (roi.py:175-222)
    tpl_pad[sy, sx] = (tpl - float(np.nanmean(tpl))) / (float(np.nanstd(tpl)) + eps)
    F = fft2d_stack(np.stack([img_z.astype(np.float32), tpl_pad]))                  # fftshift-ed complex64 spectra from the device
    F = np.fft.ifftshift(F, axes=(-2, -1)).astype(np.complex128)
    prod = F[0] * np.conj(F[1])
    prod /= np.maximum(np.abs(prod), 100 * np.finfo(np.float64).eps)               # normalization="phase"
    cc = np.fft.ifft2(prod)
    maxima = np.unravel_index(int(np.argmax(np.abs(cc))), cc.shape)
    # the shift arithmetic runs in the precision the reference's call would use: float32 frames give complex64 spectra there,
    # so its 0.1-px grid values are float32 numbers (3.3 comes back as 3.299999952316284)
    ft = np.float32 if img_z.dtype == np.float32 else np.float64
    shape = np.array([H, W])
    shift = np.array(maxima, dtype=ft)
    mid = np.fix(shape / 2)
    shift[shift > mid] -= shape[shift > mid]
    if subpixel:
        uf = ft(10)
        shift = np.round(shift * uf) / uf
        region = np.ceil(uf * ft(1.5))
        dftshift = np.fix(region / ft(2))
        up = np.conj(_upsampled_dft(np.conj(prod), int(region), 10.0, (dftshift - shift * uf).astype(np.float64)))
        m2 = np.unravel_index(int(np.argmax(np.abs(up))), up.shape)
        shift = shift + (np.array(m2, dtype=ft) - dftshift) / uf
    shift[shape == 1] = 0
    return float(shift[0]), float(shift[1]), float("nan"), float("nan")


@_register("phase")
def phase_correlation(template, image, *, slices_yx=None, backend: Literal["internal", "skimage"] = "internal",
                      subpixel: bool = True, eps: float = 1e-9):
    """Translation (dy, dx) of a template ROI inside a full frame by phase correlation
    (reference: tracking.py:191-297).  Returns (dy, dx, peak_value, snr) as Python floats."""
    torch = _ffi.require_gpu()
    tpl = _as_float2d(template, name="template")
    img = _as_float2d(image, name="image")
    H, W = img.shape
    h, w = tpl.shape
    if slices_yx is None:
        slices_yx = roi_slices((H, W), (h, w), center_yx=None, clip=False)
    sy, sx = slices_yx
    if (sy.stop - sy.start, sx.stop - sx.start) != (h, w):
        raise ValueError("ROI shape does not match target slice dimensions.")
    if backend == "skimage":
        return _phase_correlation_upsampled(tpl, img, (sy, sx), subpixel, eps)
    if backend != "internal":
        raise ValueError("backend must be 'internal' or 'skimage'.")
    timg, _, _ = D.to_device_f32(img[None], ndim=(3,))
    ttpl, _, _ = D.to_device_f32(tpl, ndim=(2,))
    canvas = torch.zeros((1, H, W), dtype=torch.float32, device=timg.device)
    canvas[0, sy, sx] = ttpl
    r = phase_correlation_batch(timg, canvas, [0], [[sy.start, sy.stop, sx.start, sx.stop]], [0], [0],
                                subpixel=subpixel, eps=eps)[0]
    return float(r[0]), float(r[1]), float(r[2]), float(r[3])
