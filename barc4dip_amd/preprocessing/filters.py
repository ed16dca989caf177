"""Gaussian-PSF deconvolution on the GPU -- drop-in for ``barc4dip.preprocessing.filters.deconvolve_psf``.

Same signature, defaults and error behaviour as the reference (filters.py:17-191).  ``method="wiener"`` runs the
whole per-frame chain on the device (reflect pad, max-abs normalisation, Wiener-Hunt filter in the Fourier domain of
the padded size, clip, rescale, crop: b4d_wiener_*).  The reference delegates the filter to
``skimage.restoration.wiener``; that library cannot be installed here, so the filter follows its published
definition (Laplacian regulariser, `balance`) and parity is UNPINNED.  ``method="rl"`` runs
Richardson-Lucy as published for ``skimage.restoration.richardson_lucy`` (``b4d_richardson_lucy``: two LDS-tiled
direct convolutions per iteration, same padding / normalisation / crop; parity likewise unpinned).  ``method="uw"``
runs scikit-image's unsupervised Wiener-Hunt Gibbs sampler as published (``b4d_uw_step``: one fused pass over the
half-plane spectrum per sweep; transforms through the library's own fft2d / ifft2d; the two Gamma draws per sweep on the
host).  It is stochastic by construction -- the reference passes no ``rng``, two of its own runs differ -- so the extra
keyword ``rng`` exists here: ``None`` draws the normals on the device (Philox), an int / ``numpy.random.Generator`` replays
the library's host stream (two normal fields, two Gamma variates per sweep), which is what the tests compare.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from typing import Literal, Sequence

import numpy as np

from .. import _device as D
from .. import _ffi

logger = logging.getLogger(__name__)
_DeconvMethod = Literal["wiener", "rl", "uw"]


def _parse_sigma(sigma) -> tuple[float, float]:
    if isinstance(sigma, (int, float, np.floating)):
        sy = sx = float(sigma)
    else:
        vals = list(sigma)
        if len(vals) != 2:
            raise ValueError("sigma must be a float or a length-2 sequence (sy, sx).")
        sy, sx = float(vals[0]), float(vals[1])
    if not (np.isfinite(sy) and np.isfinite(sx)):
        raise ValueError("sigma values must be finite.")
    if sy <= 0 or sx <= 0:
        raise ValueError("sigma values must be > 0.")
    return sy, sx


def _odd(n: int) -> int:
    n = int(n)
    return n if n % 2 == 1 else n + 1


def _gaussian_psf(sy: float, sx: float, *, min_size: int = 5) -> np.ndarray:
    """Normalised float32 Gaussian on an odd support of max(min_size, ceil(6 sigma)) samples (filters.py:217-230)."""
    ky = _odd(max(min_size, int(np.ceil(6.0 * sy))))
    kx = _odd(max(min_size, int(np.ceil(6.0 * sx))))
    yy = (np.arange(ky, dtype=np.float32) - (ky - 1) / 2.0)[:, None]
    xx = (np.arange(kx, dtype=np.float32) - (kx - 1) / 2.0)[None, :]
    psf = np.exp(-0.5 * ((yy / sy) ** 2 + (xx / sx) ** 2)).astype(np.float32, copy=False)
    total = float(psf.sum())
    if not np.isfinite(total) or total <= 0:
        raise ValueError("Failed to build a valid Gaussian PSF (sum<=0).")
    psf /= total
    return psf


class _WienerPlan:
    def __init__(self, h, w, psf, balance):
        _ffi.require_gpu()
        self._h = C.c_void_p()
        p = np.ascontiguousarray(psf, dtype=np.float32)
        _ffi.check(_ffi.lib().b4d_wiener_create(int(h), int(w), p.ctypes.data_as(C.c_void_p), int(p.shape[0]), int(p.shape[1]),
                                                float(balance), C.byref(self._h)))

    def apply(self, frames, clip: bool):
        import torch

        out = torch.empty_like(frames)
        _ffi.check(_ffi.lib().b4d_wiener_apply(self._h, D.ptr(frames), int(frames.shape[0]), D.ptr(out), int(bool(clip)),
                                               _ffi.stream_ptr()))
        return out

    def __del__(self):
        try:
            if self._h:
                _ffi.lib().b4d_wiener_destroy(self._h)
        except Exception:
            pass


def _uw_one_frame(frame, psf: np.ndarray, clip: bool, reg, user_params, rng):
    """method='uw' for one device frame (H, W) float32 -> restored device frame (filters.py:252-289 around
    skimage.restoration.unsupervised_wiener, published algorithm; see csrc/b4d_uw.hip for the sweep)."""
    import torch

    from ..signal import fft as sfft

    params = {"threshold": 1e-4, "max_num_iter": 200, "min_num_iter": 30, "burnin": 15, "callback": None}
    params.update(user_params or {})
    py, px = int(psf.shape[0] // 2), int(psf.shape[1] // 2)
    padded = torch.nn.functional.pad(frame[None, None], (px, px, py, py), mode="reflect")[0, 0]
    mag = padded.abs()
    finite = mag[~torch.isnan(mag)]                       # np.nanmax: NaNs are skipped, an all-NaN frame has no scale
    scale = float(finite.max()) if finite.numel() else float("nan")
    if not np.isfinite(scale) or scale == 0.0:
        return torch.zeros_like(frame)
    work = (padded / scale).to(torch.float32).contiguous()
    ny, nx = (int(v) for v in work.shape)
    nxh, npix = nx // 2 + 1, ny * nx

    def half_spectrum(img2d):                             # rfft2 layout out of the library's shifted full spectrum
        F = sfft.fft2d_stack(img2d[None], return_tensors=True)[0]
        return torch.roll(F, shifts=(-(ny // 2), -(nx // 2)), dims=(0, 1))[:, :nxh].contiguous()

    def ir2tf(kernel):                                    # kernel centred on the origin of an image-sized array
        big = np.zeros((ny, nx), dtype=np.float32)
        big[:kernel.shape[0], :kernel.shape[1]] = kernel
        for ax, k in enumerate(kernel.shape):
            big = np.roll(big, -int(np.floor(k / 2)), axis=ax)
        return half_spectrum(torch.from_numpy(big).to(work.device))

    if reg is None:
        lap = np.zeros((3, 3), dtype=np.float32)
        lap[1, 1] = 4.0
        lap[0, 1] = lap[2, 1] = lap[1, 0] = lap[1, 2] = -1.0
        L = ir2tf(lap)
    elif np.iscomplexobj(reg):
        L = torch.from_numpy(np.ascontiguousarray(reg, dtype=np.complex64)).to(work.device)
        if tuple(L.shape) != (ny, nxh):
            raise ValueError(f"reg: a transfer function must have the half-plane shape {(ny, nxh)} of the padded frame")
    else:
        rk = np.asarray(reg, dtype=np.float32)
        if rk.ndim != 2 or rk.shape[0] > ny or rk.shape[1] > nx:
            raise ValueError("reg must be None, a 2-D impulse response no larger than the padded frame, or a complex transfer function")
        L = ir2tf(rk)
    H = ir2tf(np.asarray(psf, dtype=np.float32))
    areg2 = (L.real * L.real + L.imag * L.imag).contiguous()
    Y = (half_spectrum(work) / float(np.sqrt(npix))).contiguous()          # unitary transform
    post = torch.zeros_like(Y)
    xs = torch.empty_like(Y) if params["callback"] else None
    sums = torch.zeros(4, dtype=torch.float64, device=work.device)
    host_normals = rng is not None
    gen = np.random.default_rng(rng)
    seed = 0 if host_normals else int.from_bytes(os.urandom(8), "little")   # like the reference: a fresh stream per call
    lib = _ffi.lib()
    gn, gx = [1.0], [1.0]
    burn, delta, it = int(params["burnin"]), float("nan"), 0
    for it in range(int(params["max_num_iter"])):
        r1 = r2 = None
        if host_normals:                                   # the library's order of draws: two normal fields, then two Gamma variates
            r1 = torch.from_numpy(gen.standard_normal((ny, nxh)).astype(np.float32)).to(work.device)
            r2 = torch.from_numpy(gen.standard_normal((ny, nxh)).astype(np.float32)).to(work.device)
        _ffi.check(lib.b4d_uw_step(D.ptr(Y), D.ptr(H), D.ptr(areg2), D.ptr(xs) if xs is not None else None, D.ptr(post),
                                   D.ptr(r1) if r1 is not None else None, D.ptr(r2) if r2 is not None else None, seed, it, burn,
                                   float(gn[-1]), float(gx[-1]), ny, nxh, D.ptr(sums), _ffi.stream_ptr()))
        q1, q2, d1, d2 = (float(v) for v in sums.cpu().numpy())
        if params["callback"]:
            params["callback"](xs.cpu().numpy())
        gn.append(gen.gamma(npix / 2, 2 / q1))
        gx.append(gen.gamma((npix - 1) / 2, 2 / q2))
        if it > burn + 1:
            delta = d1 / d2 / (it - burn)
        if it > params["min_num_iter"] and delta < params["threshold"]:
            break
    post = post / float(it - burn)
    # full Hermitian spectrum -> the library's ifft2d (fftshift-ed input); Re(ifft2) of it is irfft2 of the half plane
    full = torch.empty((ny, nx), dtype=post.dtype, device=post.device)
    full[:, :nxh] = post
    if nx - nxh > 0:
        ky = (-torch.arange(ny, device=post.device)) % ny
        kx = nx - torch.arange(nxh, nx, device=post.device)
        full[:, nxh:] = torch.conj(post[ky][:, kx])
    full = torch.roll(full, shifts=(ny // 2, nx // 2), dims=(0, 1)).contiguous()
    out = sfft.ifft2d(full, return_tensors=True).real * float(np.sqrt(npix))   # unitary inverse
    if clip:
        out = out.clamp(-1.0, 1.0)
    out = (out.to(torch.float32) * scale)[py:ny - py, px:nx - px]
    return out.contiguous()


_wiener_plans: dict = {}


def _wiener_plan(h, w, psf, balance) -> _WienerPlan:
    """Plans (twiddles, transposed filter, work buffers) are cached per (shape, PSF, balance, device, stream): the work
    buffers belong to the calls queued on ONE stream."""
    import torch

    key = (int(h), int(w), psf.shape, psf.tobytes(), float(balance), torch.cuda.current_device(),
           int(torch.cuda.current_stream().cuda_stream))
    pl = _wiener_plans.get(key)
    if pl is None:
        if len(_wiener_plans) >= 4:
            _wiener_plans.pop(next(iter(_wiener_plans)))
        pl = _wiener_plans[key] = _WienerPlan(h, w, psf, balance)
    return pl


def deconvolve_psf(images: np.ndarray, *, sigma: float | Sequence[float], method: _DeconvMethod = "wiener", clip: bool = True,
                   pad_mode: Literal["reflect"] = "reflect", balance: float | None = None, num_iter: int = 50,
                   filter_epsilon: float | None = None, reg: float | None = None, user_params: dict | None = None,
                   is_real: bool = True, parallel: bool = True, n_jobs: int | None = None, verbose: bool = False,
                   return_tensors: bool = False, rng=None) -> np.ndarray:
    """Deconvolve a 2-D image or (T, H, W) stack with a Gaussian PSF of std `sigma` (pixels).  Returns float32 of the
    input shape.  `parallel` / `n_jobs` are accepted for signature compatibility (frames are batched on the device)."""
    if not isinstance(images, np.ndarray) and not D.is_tensor(images):
        raise TypeError("deconvolve_psf expects a numpy.ndarray")
    if images.ndim not in (2, 3):
        raise ValueError(f"images must be 2D (H, W) or 3D (T, H, W); got ndim={images.ndim}")
    sy, sx = _parse_sigma(sigma)
    psf = _gaussian_psf(sy, sx, min_size=5)
    if method not in {"wiener", "rl", "uw"}:
        raise ValueError(f"Unsupported method: {method!r}. Use 'wiener', 'rl', or 'uw'.")
    if pad_mode != "reflect":
        raise ValueError("Only pad_mode='reflect' is supported (by design).")
    if method == "uw" and not is_real:
        raise NotImplementedError("method='uw' with is_real=False (full-plane sampler, complex result) is not built; the default is_real=True is.")
    if balance is None:
        balance = 0.01
    stack = images if images.ndim == 3 else images[None]
    dev, _, _ = D.to_device_f32(stack, ndim=(3,))
    if method == "uw":
        import torch

        gen = np.random.default_rng(rng) if rng is not None else None     # ONE stream over the frames, in frame order
        out = torch.stack([_uw_one_frame(dev[t], psf, bool(clip), reg, user_params, gen) for t in range(int(dev.shape[0]))])
    elif method == "rl":
        if num_iter < 1:
            raise ValueError("num_iter must be >= 1 for method='rl'.")
        import torch

        out = torch.empty_like(dev)
        p32 = np.ascontiguousarray(psf, dtype=np.float32)
        _ffi.check(_ffi.lib().b4d_richardson_lucy(D.ptr(dev), int(dev.shape[0]), int(dev.shape[1]), int(dev.shape[2]),
                                                  p32.ctypes.data_as(C.c_void_p), int(p32.shape[0]), int(p32.shape[1]), int(num_iter),
                                                  float(filter_epsilon or 0.0), int(bool(clip)), D.ptr(out), _ffi.stream_ptr()))
    else:
        plan = _wiener_plan(dev.shape[1], dev.shape[2], psf, balance)
        out = plan.apply(dev, clip)
    if images.ndim == 2:
        out = out[0]
    if verbose:
        logger.info("> deconvolve_psf | frames=%d | method=%s | sigma=(%.3f, %.3f) px | kernel=%dx%d | device batch",
                    int(stack.shape[0]), method, sy, sx, int(psf.shape[0]), int(psf.shape[1]))
    return out if return_tensors else D.to_host(out, np.float32)
