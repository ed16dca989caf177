"""Thin device wrappers around the reduction kernels of libb4d (b4d_stats.hip).

Everything here takes/returns ROCm tensors; the reference-shaped functions in statistics.py,
sharpness.py, speckles.py and temporal.py are built on these.
"""
from __future__ import annotations

import numpy as np

from .. import _device as D
from .. import _ffi


def moments_batch(frames, *, eps: float = 1e-6, saturation: float | None = 65535.0):
    """Per-frame finite-only sums.  frames: (B, ...) -> (B, 8) float64 tensor
    {n_finite, mean, sum d^2, sum d^3, sum d^4, n_zero, n_sat, 0}."""
    torch = _ffi.require_gpu()
    t, _, _ = D.to_device_f32(frames, ndim=(2, 3))
    b = int(t.shape[0])
    npix = int(t[0].numel())
    if npix % 4:
        raise NotImplementedError("frame pixel count must be a multiple of 4")
    out = torch.empty((b, 8), dtype=torch.float64, device=t.device)
    sat = float("inf") if saturation is None else float(saturation)
    _ffi.check(_ffi.lib().b4d_moments(D.ptr(t), b, npix, float(eps), sat, D.ptr(out), _ffi.stream_ptr()))
    return out


def sobel_laplace_batch(frames):
    """frames (B, ny, nx) -> (B, 4) float64 tensor {mean gx^2, mean gy^2, mean lap, mean lap^2} over finite pixels
    (scipy.ndimage sobel / laplace, mode='reflect')."""
    torch = _ffi.require_gpu()
    t, _, _ = D.to_device_f32(frames, ndim=(3,))
    b, ny, nx = (int(v) for v in t.shape)
    out = torch.empty((b, 4), dtype=torch.float64, device=t.device)
    _ffi.check(_ffi.lib().b4d_sobel_laplace_stats(D.ptr(t), b, ny, nx, D.ptr(out), _ffi.stream_ptr()))
    return out


def temporal_accumulate(frames, sum_x, sum_xx):
    """sum_x += sum_t frames, sum_xx += sum_t frames^2 (float64 device accumulators, in place)."""
    t, _, _ = D.to_device_f32(frames, ndim=(3,))
    npix = int(t[0].numel())
    _ffi.check(_ffi.lib().b4d_temporal_accumulate(D.ptr(t), int(t.shape[0]), npix, D.ptr(sum_x), D.ptr(sum_xx),
                                                  _ffi.stream_ptr()))


def temporal_finalize(sum_x, sum_xx, count: float):
    """(mean, var, contrast) float32 maps shaped like sum_x from the (all-reduced) sums."""
    torch = _ffi.require_gpu()
    mean = torch.empty(sum_x.shape, dtype=torch.float32, device=sum_x.device)
    var = torch.empty_like(mean)
    con = torch.empty_like(mean)
    _ffi.check(_ffi.lib().b4d_temporal_finalize(D.ptr(sum_x), D.ptr(sum_xx), float(count), int(sum_x.numel()),
                                                D.ptr(mean), D.ptr(var), D.ptr(con), _ffi.stream_ptr()))
    return mean, var, con


def percentiles_batch(frames, q):
    """np.nanpercentile(frame, q) (linear interpolation) for every frame: (B, ...) -> (B, len(q)) float64 ndarray."""
    import ctypes as C

    import numpy as np

    torch = _ffi.require_gpu()
    t, _, _ = D.to_device_f32(frames, ndim=(2, 3))
    b = int(t.shape[0])
    npix = int(t[0].numel())
    qs = np.ascontiguousarray(q, dtype=np.float64).ravel()
    out = torch.empty((b, qs.size, 4), dtype=torch.float64, device=t.device)
    _ffi.check(_ffi.lib().b4d_percentiles(D.ptr(t), b, npix, qs.ctypes.data_as(C.c_void_p), int(qs.size), D.ptr(out),
                                          _ffi.stream_ptr()))
    r = out.cpu().numpy()
    lo, hi, n = r[..., 0], r[..., 1], r[..., 3]
    # NumPy's virtual index for method="linear" (alpha = beta = 1), same expression order as
    # numpy.lib._function_base_impl._compute_virtual_index
    qf = np.true_divide(qs, 100)[None, :]
    vi = n * qf + (1.0 + qf * (1.0 - 1.0 - 1.0)) - 1.0
    frac = vi - np.floor(vi)
    diff = hi - lo
    # NumPy's _lerp: a + diff*t, but b - diff*(1-t) for t >= 0.5
    return np.where(frac >= 0.5, hi - diff * (1.0 - frac), lo + diff * frac)


def psd_stats_batch(psd):
    """(B, ny, nx) shifted PSD maps (device) -> (B, 8) float64 ndarray
    {S_disc, sum FR^2 P, sum FX^2 P, sum FY^2 P, sum P^2, S_all, sum P ln P, f95}."""
    torch = _ffi.require_gpu()
    t, _, _ = D.to_device_f32(psd, ndim=(3,))
    b, ny, nx = (int(v) for v in t.shape)
    out = torch.empty((b, 8), dtype=torch.float64, device=t.device)
    _ffi.check(_ffi.lib().b4d_psd_stats(D.ptr(t), b, ny, nx, D.ptr(out), _ffi.stream_ptr()))
    return out.cpu().numpy()
