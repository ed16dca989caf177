"""FFT-based circular correlation on the GPU -- drop-in for ``barc4dip.signal.corr``.

Same conventions as the reference (signal/corr.py:1-31): circular correlation via FFT, zero
lag at the centre (fftshift), ``normalize`` in {"none", "peak"}, optional mean removal and
standardisation, lag axes in pixels or calibrated units.

The autocorrelation is computed as ONE real forward transform, |F|^2 with the DC bin zeroed
(= mean removal) and one inverse transform (the reference runs three complex128 FFTs,
signal/corr.py:237-240); results are float32-accurate and returned as float64 arrays like the
reference's.

``xcorr2d`` return type (signal/corr.py:41-42, 242): the reference passes its complex128 result through
``np.real_if_close(tol=1000)``, i.e. it returns float64 only when the rounding noise left in the imaginary
part by its three complex128 transforms stays under 1000 eps = 2.2e-13 ABSOLUTE, and complex128 otherwise --
which is the normal case for detector data (|corr| ~ 1e12).  The imaginary part of a correlation of real
inputs is exactly zero; the device runs real (half-spectrum) transforms and never forms that noise.  So this
module returns complex128 with an exactly-zero imaginary part where the reference's noise model
(|Im| ~ 0.25 eps max|corr| before the peak normalisation, calibrated on tests/golden/signal_small.npz:
observed 0.1 ... 0.45 eps max|corr|) says the reference would: max|corr| > 4000.  Within a factor ~2 of that
bound the reference's own dtype depends on the rounding of its data; everywhere else (raw detector counts:
complex128; standardised small tiles: float64) the dtypes agree.  ``np.argmax`` of the complex result orders by
(real, imag) and therefore finds the same element as on the real part.
"""
from __future__ import annotations

from typing import Literal

import numpy as np

from .. import _device as D
from .. import _ffi
from .common import _lag_axis_from_step, _resolve_step_1d, _resolve_steps_2d


def _as_real_if_close(z: np.ndarray) -> np.ndarray:
    return np.real_if_close(z, tol=1000)


# modelled size of the reference's imaginary rounding noise in xcorr2d, in units of eps * max|corr| (module docstring)
_REAL_IF_CLOSE_NOISE = 0.25


def xcorr1d(a, b, *, x=None, dx: float = 1.0, remove_mean: bool = True, standardize: bool = False,
            normalize: Literal["none", "peak"] = "peak"):
    """Circular cross-correlation of two 1-D signals (reference: signal/corr.py:45-121).  Host-side."""
    aa = np.asarray(a, dtype=float)
    bb = np.asarray(b, dtype=float)
    if aa.ndim != 1 or bb.ndim != 1:
        raise ValueError("a and b must be 1D arrays.")
    if aa.size != bb.size:
        raise ValueError("a and b must have the same length.")
    n = int(aa.size)
    lag = _lag_axis_from_step(n, _resolve_step_1d(n=n, x=x, dx=dx, name="x"))
    if remove_mean:
        aa = aa - float(np.mean(aa))
        bb = bb - float(np.mean(bb))
    if standardize:
        sa, sb = float(np.std(aa)), float(np.std(bb))
        aa = aa / sa if sa > 0 else aa
        bb = bb / sb if sb > 0 else bb
    Fa = np.fft.fft(aa)
    Fb = np.fft.fft(bb)
    corr = _as_real_if_close(np.fft.fftshift(np.fft.ifft(Fa * np.conjugate(Fb))))
    if normalize == "none":
        return corr, lag
    if normalize == "peak":
        m = float(np.max(np.abs(corr)))
        return (corr / m if m > 0 else corr), lag
    raise ValueError(f"Invalid normalize='{normalize}'. Use 'none' or 'peak'.")


def autocorr1d(a, *, x=None, dx: float = 1.0, remove_mean: bool = True, standardize: bool = False,
               normalize: Literal["none", "peak"] = "peak"):
    """Circular auto-correlation of a 1-D signal (reference: signal/corr.py:124-166).  Host-side."""
    return xcorr1d(a, a, x=x, dx=dx, remove_mean=remove_mean, standardize=standardize, normalize=normalize)


def _flags(remove_mean: bool, normalize: str) -> int:
    if normalize not in ("none", "peak"):
        raise ValueError(f"Invalid normalize='{normalize}'. Use 'none' or 'peak'.")
    return (_ffi.REMOVE_MEAN if remove_mean else 0) | (_ffi.NORM_PEAK if normalize == "peak" else 0)


def _frame_variances(t):
    """Population variance (ddof 0) of every (B, ny, nx) frame from the float64 power sums of b4d_moments -> (B,) float64
    device tensor (corr.py:229-235 standardises by np.std)."""
    from ..metrics import kernels as K
    from ..metrics.speckles import _pad4

    mom = K.moments_batch(_pad4(t), eps=0.0, saturation=None)        # {n, mean, sum d^2, ...}
    return mom[:, 2] / mom[:, 0]


def _std_scale(t, remove_mean: bool):
    """1/var per frame (population variance, corr.py:229-235); 1 where the variance is 0."""
    var = _frame_variances(t)
    one = var.new_ones(())
    return (one / var.where(var > 0, one)).float()


def autocorr2d_stack(stack, *, remove_mean: bool = True, standardize: bool = False,
                     normalize: Literal["none", "peak"] = "peak", return_tensors: bool = False):
    """Batched autocorr2d over a (T, ny, nx) stack -> (T, ny, nx) float32 (device arithmetic)."""
    torch = _ffi.require_gpu()
    t, _, _ = D.to_device_f32(stack, ndim=(3,))
    T, ny, nx = t.shape
    flags = _flags(remove_mean, normalize)
    pl = _ffi.get_plan(ny, nx, _ffi.stack_chunk(ny, nx, int(T)))
    out = torch.empty((T, ny, nx), dtype=torch.float32, device=t.device)
    _ffi.check(_ffi.lib().b4d_autocorr2d(pl.handle, D.ptr(t), int(T), D.ptr(out), flags, _ffi.stream_ptr()))
    if standardize and normalize == "none":
        out *= _std_scale(t, remove_mean)[:, None, None]
    return out if return_tensors else D.to_host(out, np.float64)


def psd_autocorr2d_stack(stack, *, dx: float = 1.0, dy: float = 1.0, scale: bool = True, remove_mean: bool = True,
                         normalize: Literal["none", "peak"] = "peak", return_tensors: bool = False,
                         out_psd=None, out_autocorr=None):
    """The fused north-star pipeline: one forward transform per frame serves psd2d AND autocorr2d.

    Returns (psd (T, ny, nx) float32, autocorr (T, ny, nx) float32).  `out_*` let a caller reuse
    device buffers (benchmarks)."""
    torch = _ffi.require_gpu()
    t, _, _ = D.to_device_f32(stack, ndim=(3,))
    T, ny, nx = t.shape
    flags = _flags(remove_mean, normalize)
    pl = _ffi.get_plan(ny, nx, _ffi.stack_chunk(ny, nx, int(T)))
    psd = out_psd if out_psd is not None else torch.empty((T, ny, nx), dtype=torch.float32, device=t.device)
    ac = out_autocorr if out_autocorr is not None else torch.empty((T, ny, nx), dtype=torch.float32, device=t.device)
    s = (dx * dy) / (float(nx) * float(ny)) if scale else 1.0
    _ffi.check(_ffi.lib().b4d_psd_autocorr2d(pl.handle, D.ptr(t), int(T), D.ptr(psd), float(s), D.ptr(ac), flags,
                                             _ffi.stream_ptr()))
    if return_tensors:
        return psd, ac
    return D.to_host(psd), D.to_host(ac)


def xcorr2d(a, b, *, x=None, y=None, dx: float = 1.0, dy: float = 1.0, remove_mean: bool = True,
            standardize: bool = False, normalize: Literal["none", "peak"] = "peak", return_tensors: bool = False):
    """Circular cross-correlation of two 2-D signals (reference: signal/corr.py:169-253).

    Returns (corr (ny, nx) float64, or complex128 with a zero imaginary part where the reference's
    ``np.real_if_close`` would keep its rounding-noise imaginary part -- see the module docstring --, xlag (nx,),
    ylag (ny,))."""
    torch = _ffi.require_gpu()
    if not D.is_tensor(a):
        a = np.asarray(a)
    if not D.is_tensor(b):
        b = np.asarray(b)
    if a.ndim != 2 or b.ndim != 2:
        raise ValueError("a and b must be 2D arrays.")
    if tuple(a.shape) != tuple(b.shape):
        raise ValueError("a and b must have the same shape.")
    ny, nx = a.shape
    sx, sy = _resolve_steps_2d(shape=(ny, nx), x=x, y=y, dx=dx, dy=dy)
    xlag, ylag = _lag_axis_from_step(nx, sx), _lag_axis_from_step(ny, sy)
    _flags(remove_mean, normalize)            # validates `normalize`
    flags = _ffi.REMOVE_MEAN if remove_mean else 0
    ta, _, _ = D.to_device_f32(a[None], ndim=(3,))
    tb, _, _ = D.to_device_f32(b[None], ndim=(3,))
    pl = _ffi.get_plan(ny, nx)
    out = torch.empty((1, ny, nx), dtype=torch.float32, device=ta.device)
    _ffi.check(_ffi.lib().b4d_xcorr2d(pl.handle, D.ptr(ta), D.ptr(tb), 1, D.ptr(out), flags, _ffi.stream_ptr()))
    if standardize:
        sa = float(_frame_variances(ta)[0]) ** 0.5
        sb = float(_frame_variances(tb)[0]) ** 0.5
        out *= float(1.0 / ((sa if sa > 0 else 1.0) * (sb if sb > 0 else 1.0)))
    corr = out[0]
    m = float(corr.abs().max())               # max|corr| as the reference sees it at np.real_if_close (corr.py:242)
    if return_tensors:                        # extension: the real correlation as a device tensor
        return (corr / m if normalize == "peak" and m > 0 else corr), xlag, ylag
    host = D.to_host(corr, np.float64)
    if _REAL_IF_CLOSE_NOISE * np.finfo(np.float64).eps * m > 1000 * np.finfo(np.float64).eps or not np.isfinite(m):
        host = host.astype(np.complex128)     # the reference's imaginary part is rounding noise; ours is exactly zero
    if normalize == "peak" and m > 0:
        host = host / m                       # corr.py:247-250, in float64 like the reference
    return host, xlag, ylag


def autocorr2d(a, *, x=None, y=None, dx: float = 1.0, dy: float = 1.0, remove_mean: bool = True,
               standardize: bool = False, normalize: Literal["none", "peak"] = "peak",
               return_tensors: bool = False):
    """Circular auto-correlation of a 2-D signal (reference: signal/corr.py:256-320).

    Returns (corr (ny, nx) float64 with the peak exactly 1.0 at [ny//2, nx//2] when
    normalize="peak", xlag, ylag)."""
    if not D.is_tensor(a):
        a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError("a and b must be 2D arrays.")
    ny, nx = a.shape
    sx, sy = _resolve_steps_2d(shape=(ny, nx), x=x, y=y, dx=dx, dy=dy)
    xlag, ylag = _lag_axis_from_step(nx, sx), _lag_axis_from_step(ny, sy)
    ac = autocorr2d_stack(a[None], remove_mean=remove_mean, standardize=standardize, normalize=normalize,
                          return_tensors=True)[0]
    return (ac if return_tensors else D.to_host(ac, np.float64)), xlag, ylag
