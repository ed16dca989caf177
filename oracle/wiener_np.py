"""Oracle (test infrastructure, NOT product code): Wiener path of
``barc4dip.preprocessing.deconvolve_psf``.

PARITY UNPINNED.  The arithmetic of the reference lives in
``skimage.restoration.wiener`` (preprocessing/filters.py:266); scikit-image is
listed un-pinned in the reference's pyproject.toml:19-28 and is absent from this
image, and the reference holds no tests or vectors for this path.  What is
restated here:

* pad / normalise / rescale / crop, Gaussian PSF, sigma parsing:
  preprocessing/filters.py:194-289 (read as text);
* the Wiener-Hunt filter as published for ``skimage.restoration.wiener``
  (Orieux, Giovannelli, Rodet, JOSA A 27(7), 2010; scikit-image API docs):
  ``reg`` = transfer function of the 2-D discrete Laplacian, ``ir2tf`` = zero-pad
  the kernel to the image shape, roll every axis by -floor(size/2), real FFT;
  ``W = conj(H) / (|H|^2 + balance*|L|^2)``; ``out = irfft2(W * rfft2(img))``;
  ``clip`` to [-1, 1]; float32 input stays in float32 arithmetic.

Self-checks (tests/test_wiener_oracle.py) replace parity: balance->0 inverse
limit on a band-limited image, agreement with an independent float64 full-FFT
formulation, blur->deconvolve error reduction.
"""
from __future__ import annotations

import numpy as np


def parse_sigma(sigma):
    """preprocessing/filters.py:194-209."""
    if isinstance(sigma, (int, float, np.floating)):
        sy = sx = float(sigma)
    else:
        s = list(sigma)
        if len(s) != 2:
            raise ValueError("sigma must be a float or a length-2 sequence (sy, sx).")
        sy, sx = float(s[0]), float(s[1])
    if not (np.isfinite(sy) and np.isfinite(sx)):
        raise ValueError("sigma values must be finite.")
    if sy <= 0 or sx <= 0:
        raise ValueError("sigma values must be > 0.")
    return sy, sx


def _odd(n):
    n = int(n)
    return n if n % 2 == 1 else n + 1


def gaussian_psf(sy, sx, *, min_size=5):
    """preprocessing/filters.py:217-230 -- odd(max(5, ceil(6 sigma))) support, float32, sum 1."""
    ky = _odd(max(min_size, int(np.ceil(6.0 * sy))))
    kx = _odd(max(min_size, int(np.ceil(6.0 * sx))))
    y = np.arange(ky, dtype=np.float32) - (ky - 1) / 2.0
    x = np.arange(kx, dtype=np.float32) - (kx - 1) / 2.0
    yy, xx = np.meshgrid(y, x, indexing="ij")
    psf = np.exp(-0.5 * ((yy / sy) ** 2 + (xx / sx) ** 2)).astype(np.float32, copy=False)
    s = float(psf.sum())
    if not np.isfinite(s) or s <= 0:
        raise ValueError("Failed to build a valid Gaussian PSF (sum<=0).")
    return psf / np.float32(s)


def ir2tf(kernel, shape, dtype):
    """Published skimage.restoration.uft.ir2tf (real=True): centred kernel -> rfft2 transfer function."""
    big = np.zeros(shape, dtype=dtype)
    big[tuple(slice(0, s) for s in kernel.shape)] = kernel
    for ax, s in enumerate(kernel.shape):
        big = np.roll(big, shift=-int(np.floor(s / 2)), axis=ax)
    return np.fft.rfft2(big)


def laplacian_tf(shape, dtype):
    """Published skimage.restoration.uft.laplacian for ndim=2: kernel [[0,-1,0],[-1,4,-1],[0,-1,0]]."""
    k = np.zeros((3, 3), dtype=dtype)
    k[1, 1] = 4.0
    k[0, 1] = k[2, 1] = k[1, 0] = k[1, 2] = -1.0
    return ir2tf(k, shape, dtype)


def wiener_filter_tf(psf, shape, balance, dtype=np.float32):
    """W = conj(H) / (|H|^2 + balance |L|^2), half-spectrum (rfft2 layout)."""
    H = ir2tf(np.asarray(psf, dtype=dtype), shape, dtype)
    L = laplacian_tf(shape, dtype)
    return np.conj(H) / (np.abs(H) ** 2 + dtype(balance) * np.abs(L) ** 2)


def wiener(image, psf, balance, clip=True):
    """Published skimage.restoration.wiener(image, psf, balance, clip=True) for real images."""
    img = np.asarray(image)
    dt = np.float32 if img.dtype == np.float32 else np.float64
    img = img.astype(dt, copy=False)
    W = wiener_filter_tf(psf, img.shape, balance, dtype=dt)
    out = np.fft.irfft2(W * np.fft.rfft2(img), s=img.shape).astype(dt, copy=False)
    if clip:
        out = np.clip(out, -1.0, 1.0)
    return out


def richardson_lucy(image, psf, num_iter=50, clip=True, filter_epsilon=None):
    """Richardson-Lucy as published for ``skimage.restoration.richardson_lucy`` (PARITY UNPINNED, see the module
    header): start from a constant 0.5 image; ``conv = convolve(est, psf, "same") + 1e-12``;
    ``rel = image / conv`` (0 where ``conv < filter_epsilon`` if given); ``est *= convolve(rel, flip(psf), "same")``;
    clip to [-1, 1].  Input float32 stays float32; zero boundary."""
    from scipy.signal import convolve

    image = np.asarray(image, dtype=np.float32)
    psf = np.asarray(psf, dtype=np.float32)
    est = np.full(image.shape, 0.5, dtype=np.float32)
    mirror = np.flip(psf)
    for _ in range(int(num_iter)):
        conv = convolve(est, psf, mode="same", method="direct") + np.float32(1e-12)
        if filter_epsilon:
            rel = np.where(conv < filter_epsilon, 0, image / conv).astype(np.float32)
        else:
            rel = image / conv
        est = est * convolve(rel, mirror, mode="same", method="direct")
    if clip:
        est = np.clip(est, -1.0, 1.0)
    return est.astype(np.float32, copy=False)


def deconv_one_frame(frame, psf, balance=0.01, clip=True, method="wiener", num_iter=50, filter_epsilon=None):
    """preprocessing/filters.py:233-289 (method='wiener' | 'rl')."""
    if frame.ndim != 2:
        raise ValueError("Internal error: frame must be 2D.")
    py, px = int(psf.shape[0] // 2), int(psf.shape[1] // 2)
    padded = np.pad(frame, ((py, py), (px, px)), mode="reflect")
    scale = float(np.nanmax(np.abs(padded)))
    if not np.isfinite(scale) or scale == 0.0:
        return np.zeros_like(padded, dtype=np.float32)[py:-py, px:-px]
    work = (padded / scale).astype(np.float32, copy=False)
    if method == "rl":
        if num_iter < 1:
            raise ValueError("num_iter must be >= 1 for method='rl'.")
        restored = richardson_lucy(work, psf, num_iter=int(num_iter), clip=bool(clip), filter_epsilon=filter_epsilon)
    else:
        restored = wiener(work, psf, float(balance), clip=bool(clip))
    return (restored.astype(np.float32, copy=False) * scale)[py:-py, px:-px]


def deconvolve_psf(images, *, sigma, method="wiener", clip=True, pad_mode="reflect", balance=None, num_iter=50,
                   filter_epsilon=None):
    """preprocessing/filters.py:17-191 (method='wiener' | 'rl', serial)."""
    if not isinstance(images, np.ndarray):
        raise TypeError("deconvolve_psf expects a numpy.ndarray")
    if images.ndim not in (2, 3):
        raise ValueError(f"images must be 2D (H, W) or 3D (T, H, W); got ndim={images.ndim}")
    sy, sx = parse_sigma(sigma)
    psf = gaussian_psf(sy, sx, min_size=5)
    if method not in ("wiener", "rl"):
        raise ValueError(f"oracle restates method='wiener' and 'rl' only (got {method!r}).")
    if pad_mode != "reflect":
        raise ValueError("Only pad_mode='reflect' is supported (by design).")
    if balance is None:
        balance = 0.01
    kw = dict(method=method, num_iter=num_iter, filter_epsilon=filter_epsilon)
    img = images.astype(np.float32, copy=False)
    if img.ndim == 2:
        return deconv_one_frame(img, psf, balance, clip, **kw).astype(np.float32, copy=False)
    out = np.empty_like(img, dtype=np.float32)
    for t in range(img.shape[0]):
        out[t] = deconv_one_frame(img[t], psf, balance, clip, **kw)
    return out
