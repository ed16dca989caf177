"""Gaussian-PSF deconvolution on the GPU -- drop-in for ``barc4dip.preprocessing.filters.deconvolve_psf``.

Same signature, defaults and error behaviour as the reference (filters.py:17-191).  ``method="wiener"`` runs the
whole per-frame chain on the device (reflect pad, max-abs normalisation, Wiener-Hunt filter in the Fourier domain of
the padded size, clip, rescale, crop: b4d_wiener_*).  The reference delegates the filter to
``skimage.restoration.wiener``; that library cannot be installed here, so the filter follows its published
definition (Laplacian regulariser, `balance`) and parity is UNPINNED.  ``method="rl"`` runs
Richardson-Lucy as published for ``skimage.restoration.richardson_lucy`` (``b4d_richardson_lucy``: two LDS-tiled
direct convolutions per iteration, same padding / normalisation / crop; parity likewise unpinned).  ``method="uw"``
(scikit-image's stochastic unsupervised Wiener-Hunt sampler) is not built.
"""
from __future__ import annotations

import ctypes as C
import logging
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
                   return_tensors: bool = False) -> np.ndarray:
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
    if method == "uw":
        raise NotImplementedError("method='uw' (scikit-image's stochastic unsupervised Wiener-Hunt sampler) is not built on the GPU path.")
    if balance is None:
        balance = 0.01
    stack = images if images.ndim == 3 else images[None]
    dev, _, _ = D.to_device_f32(stack, ndim=(3,))
    if method == "rl":
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
