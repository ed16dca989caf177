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


def image_quad_norm(x):
    """Published skimage.restoration.uft.image_quad_norm: squared l2 norm of a (possibly half-plane) spectrum.  The
    Hermitian case is recognised by a NON-SQUARE array and counts every column twice except column 0 (the Nyquist column
    is counted twice as well: the library's own convention, kept)."""
    a2 = np.abs(x) ** 2
    if x.shape[-1] != x.shape[-2]:
        return 2.0 * np.sum(np.sum(a2, axis=-1), axis=-1) - np.sum(a2[..., 0], axis=-1)
    return np.sum(np.sum(a2, axis=-1), axis=-1)


def unsupervised_wiener(image, psf, reg=None, user_params=None, is_real=True, clip=True, *, rng=None, normals=None):
    """Unsupervised Wiener-Hunt deconvolution as published for ``skimage.restoration.unsupervised_wiener`` (Orieux,
    Giovannelli, Rodet, JOSA A 27(7), 2010, Eqs. 27-31, 44; scikit-image API docs) -- PARITY UNPINNED (module header), and
    stochastic by construction: the reference calls it without ``rng`` (preprocessing/filters.py:278-286), so two runs of
    the reference itself differ.  Gibbs sampler in the unitary half-plane Fourier domain:

        precision = gn |H|^2 + gx |L|^2;  x = gn conj(H) / precision * Y + sqrt(0.5 / precision) (r1 + i r2)
        gn ~ Gamma(N / 2, 2 / ||Y - x H||^2),  gx ~ Gamma((N - 1) / 2, 2 / ||x L||^2)
        posterior mean = average of x after ``burnin``; stop when the relative change of the running mean < threshold.

    ``normals`` (test hook, not in the library): a callable ``normals(iteration, shape) -> (r1, r2)`` replacing the two
    standard-normal draws, so that another implementation can be fed the very same numbers."""
    params = {"threshold": 1e-4, "max_num_iter": 200, "min_num_iter": 30, "burnin": 15, "callback": None}
    params.update(user_params or {})
    if not is_real:
        raise ValueError("oracle restates is_real=True only")
    img = np.asarray(image)
    ft = np.float32 if img.dtype == np.float32 else np.float64
    img = img.astype(ft, copy=False)
    L = laplacian_tf(img.shape, ft) if reg is None else (reg if np.iscomplexobj(reg) else ir2tf(np.asarray(reg, dtype=ft), img.shape, ft))
    H = ir2tf(np.asarray(psf, dtype=ft), img.shape, ft)
    ct = np.complex64 if ft == np.float32 else np.complex128
    L, H = L.astype(ct, copy=False), H.astype(ct, copy=False)
    areg2, atf2 = np.abs(L) ** 2, np.abs(H) ** 2
    Y = (np.fft.rfft2(img) / np.sqrt(img.size)).astype(ct, copy=False)       # unitary transform
    post = np.zeros(H.shape, dtype=ct)
    prev = np.zeros(H.shape, dtype=ct)
    delta = np.nan
    gn, gx = [1.0], [1.0]
    rng = np.random.default_rng(rng)
    burn = int(params["burnin"])
    it = 0
    for it in range(int(params["max_num_iter"])):
        precision = (ft(gn[-1]) * atf2 + ft(gx[-1]) * areg2).astype(ft, copy=False)
        if normals is None:
            r1 = rng.standard_normal(Y.shape).astype(ft, copy=False)
            r2 = rng.standard_normal(Y.shape).astype(ft, copy=False)
        else:
            r1, r2 = normals(it, Y.shape)
        x = (ft(gn[-1]) * np.conj(H) / precision) * Y + np.sqrt(ft(0.5) / precision) * (r1 + 1j * r2)
        x = x.astype(ct, copy=False)
        if params["callback"]:
            params["callback"](x)
        gn.append(rng.gamma(img.size / 2, 2 / image_quad_norm(Y - x * H)))
        gx.append(rng.gamma((img.size - 1) / 2, 2 / image_quad_norm(x * L)))
        if it > burn:
            post = prev + x
        if it > burn + 1:
            delta = np.sum(np.abs(post / (it - burn) - prev / (it - burn - 1))) / np.sum(np.abs(post)) / (it - burn)
        prev = post
        if it > params["min_num_iter"] and delta < params["threshold"]:
            break
    post = post / (it - burn)
    out = (np.fft.irfft2(post, s=img.shape) * np.sqrt(img.size)).astype(ft, copy=False)    # unitary inverse
    if clip:
        out = np.clip(out, -1.0, 1.0)
    return out, {"noise": gn, "prior": gx}


def deconv_one_frame(frame, psf, balance=0.01, clip=True, method="wiener", num_iter=50, filter_epsilon=None, **uw):
    """preprocessing/filters.py:233-289 (method='wiener' | 'rl' | 'uw'; **uw: reg, user_params, is_real, rng, normals)."""
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
    elif method == "uw":
        restored, _ = unsupervised_wiener(work, psf, clip=bool(clip), **uw)
    else:
        restored = wiener(work, psf, float(balance), clip=bool(clip))
    return (restored.astype(np.float32, copy=False) * scale)[py:-py, px:-px]


def deconvolve_psf(images, *, sigma, method="wiener", clip=True, pad_mode="reflect", balance=None, num_iter=50,
                   filter_epsilon=None, **uw):
    """preprocessing/filters.py:17-191 (serial; **uw reaches unsupervised_wiener for method='uw')."""
    if not isinstance(images, np.ndarray):
        raise TypeError("deconvolve_psf expects a numpy.ndarray")
    if images.ndim not in (2, 3):
        raise ValueError(f"images must be 2D (H, W) or 3D (T, H, W); got ndim={images.ndim}")
    sy, sx = parse_sigma(sigma)
    psf = gaussian_psf(sy, sx, min_size=5)
    if method not in ("wiener", "rl", "uw"):
        raise ValueError(f"Unsupported method: {method!r}. Use 'wiener', 'rl', or 'uw'.")
    if pad_mode != "reflect":
        raise ValueError("Only pad_mode='reflect' is supported (by design).")
    if balance is None:
        balance = 0.01
    kw = dict(method=method, num_iter=num_iter, filter_epsilon=filter_epsilon, **(uw if method == "uw" else {}))
    img = images.astype(np.float32, copy=False)
    if img.ndim == 2:
        return deconv_one_frame(img, psf, balance, clip, **kw).astype(np.float32, copy=False)
    out = np.empty_like(img, dtype=np.float32)
    for t in range(img.shape[0]):
        out[t] = deconv_one_frame(img[t], psf, balance, clip, **kw)
    return out
