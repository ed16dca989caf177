"""FFT and power-spectral-density helpers on the GPU -- drop-in for ``barc4dip.signal.fft``.

Same names, argument meaning and error behaviour as the reference (signal/fft.py): 2-D arrays
are (ny, nx); outputs are fftshift-ed with matching shifted frequency axes; calibration is
either (dx, dy) or uniformly sampled (x, y).

Differences (documented in DESIGN.md):
  * arithmetic is float32 on the device whatever the input dtype (float64 input is cast down,
    outputs are cast back up so that dtypes match what the reference returns);
  * inputs may be ROCm torch tensors; ``return_tensors=True`` keeps results on the device;
  * 2-D transforms: power-of-two ny, nx in [64, 4096] (FFT kernels), any ny, nx <= 512 (DFT-matrix products) or sides
    up to 8192 that split as 2^k * A * B with A + B <= 128, e.g. 2560 x 2160 (fused mixed-radix transform), any other
    side up to 4096 through Bluestein's chirp-z; ``NotImplementedError`` otherwise;
  * stacks (T, ny, nx) are accepted by the ``*_stack`` functions (batched launches).
1-D helpers and axes are host-side NumPy (negligible cost).
"""
from __future__ import annotations

import numpy as np

from .. import _device as D
from .. import _ffi
from .common import _resolve_step_1d, _resolve_steps_2d


def freq_axis1d(*, n: int, x=None, dx: float = 1.0) -> np.ndarray:
    """Shifted 1-D frequency axis (reference: signal/fft.py:31-55)."""
    if n < 1:
        raise ValueError("n must be >= 1.")
    step = _resolve_step_1d(n=n, x=x, dx=dx, name="x")
    return np.fft.fftshift(np.fft.fftfreq(int(n), d=step))


def freq_axes2d(*, shape, x=None, y=None, dx: float = 1.0, dy: float = 1.0):
    """Shifted 2-D frequency axes (fx, fy) (reference: signal/fft.py:58-96)."""
    ny, nx = shape
    if ny < 1 or nx < 1:
        raise ValueError("shape must contain positive integers.")
    sx, sy = _resolve_steps_2d(shape=(ny, nx), x=x, y=y, dx=dx, dy=dy)
    return (np.fft.fftshift(np.fft.fftfreq(int(nx), d=sx)), np.fft.fftshift(np.fft.fftfreq(int(ny), d=sy)))


def fft1d(signal, *, x=None, dx: float = 1.0):
    """Shifted 1-D FFT + axis (reference: signal/fft.py:99-131).  Host-side."""
    s = np.asarray(signal)
    if s.ndim != 1:
        raise ValueError("signal must be a 1D array.")
    fx = freq_axis1d(n=int(s.size), x=x, dx=dx)
    return np.fft.fftshift(np.fft.fft(s)), fx


def ifft1d(F):
    """Inverse of fft1d (reference: signal/fft.py:134-152).  Host-side."""
    F = np.asarray(F)
    if F.ndim != 1:
        raise ValueError("F must be a 1D array.")
    return np.fft.ifft(np.fft.ifftshift(F))


def psd1d(signal, *, x=None, dx: float = 1.0, scale: bool = True):
    """Shifted 1-D PSD, scale dx/n (reference: signal/fft.py:155-195).  Host-side."""
    s = np.asarray(signal)
    if s.ndim != 1:
        raise ValueError("signal must be a 1D array.")
    n = int(s.size)
    step = _resolve_step_1d(n=n, x=x, dx=dx, name="x")
    F, fx = fft1d(s, x=x, dx=dx)
    P = np.abs(F) ** 2
    if scale:
        P = P * (step / float(n))
    return P, fx


def _frames(image, nd):
    t, was_tensor, src = D.to_device_f32(image, ndim=nd)
    return t, was_tensor, src


def _finish(t, was_tensor, return_tensors, dtype):
    if return_tensors:
        return t
    return D.to_host(t, dtype)


def fft2d_stack(stack, *, return_tensors: bool = False):
    """Batched fftshift(fft2(frame)) for a (T, ny, nx) stack -> (T, ny, nx) complex64."""
    torch = _ffi.require_gpu()
    t, _, src = _frames(stack, (3,))
    T, ny, nx = t.shape
    pl = _ffi.get_plan(ny, nx, _ffi.stack_chunk(ny, nx, int(T)))
    out = torch.empty((T, ny, nx), dtype=torch.complex64, device=t.device)
    _ffi.check(_ffi.lib().b4d_fft2d(pl.handle, D.ptr(t), int(T), D.ptr(out), _ffi.stream_ptr()))
    return _finish(out, True, return_tensors, np.complex128 if src is np.float64 else np.complex64)


def _c2c(frames, inverse: bool):
    """(B, ny, nx) complex frames through b4d_fft2d_c2c on a general-length plan -> complex64 device tensor."""
    torch = _ffi.require_gpu()
    t = frames if D.is_tensor(frames) else torch.from_numpy(np.ascontiguousarray(frames))
    t = t.to(device="cuda", dtype=torch.complex64).contiguous()
    b, ny, nx = (int(v) for v in t.shape)
    pl = _ffi.get_plan(ny, nx, general=True)
    out = torch.empty_like(t)
    _ffi.check(_ffi.lib().b4d_fft2d_c2c(pl.handle, D.ptr(torch.view_as_real(t)), b, int(bool(inverse)), D.ptr(torch.view_as_real(out)),
                                        _ffi.stream_ptr()))
    return out


def _is_complex(a) -> bool:
    return bool(a.is_complex()) if D.is_tensor(a) else np.iscomplexobj(a)


def _complex_result_dtype(a):
    name = str(getattr(a, "dtype", "complex128")).replace("torch.", "")
    return np.complex64 if name == "complex64" else np.complex128


def fft2d(image, *, x=None, y=None, dx: float = 1.0, dy: float = 1.0, return_tensors: bool = False):
    """Shifted 2-D FFT of an image and shifted frequency axes (reference: signal/fft.py:198-237).

    Returns (F (ny, nx) complex, fx (nx,), fy (ny,)).  Raises ValueError if image is not 2-D.  Complex input runs the
    complex-to-complex transform of the general-length engines (b4d_fft2d_c2c)."""
    if not D.is_tensor(image):
        image = np.asarray(image)
    if image.ndim != 2:
        raise ValueError("image must be a 2D array.")
    ny, nx = image.shape
    fx, fy = freq_axes2d(shape=(ny, nx), x=x, y=y, dx=dx, dy=dy)
    if _is_complex(image):
        F = _c2c(image[None], False)[0]
        return _finish(F, True, return_tensors, _complex_result_dtype(image)), fx, fy
    F = fft2d_stack(image[None], return_tensors=True)[0]
    src = D.result_dtype(image)
    return _finish(F, True, return_tensors, np.complex128 if src is np.float64 else np.complex64), fx, fy


def ifft2d(F, *, return_tensors: bool = False):
    """Inverse FFT from a shifted spectrum, ifft2(ifftshift(F)) (reference: signal/fft.py:240-258), on the device
    (b4d_fft2d_c2c, complex64 arithmetic; complex128 input is cast down and the result cast back up)."""
    if not D.is_tensor(F):
        F = np.asarray(F)
    if F.ndim != 2:
        raise ValueError("F must be a 2D array.")
    out = _c2c(F[None], True)[0]
    return _finish(out, True, return_tensors, _complex_result_dtype(F) if _is_complex(F) else np.complex128)


def psd2d_stack(stack, *, dx: float = 1.0, dy: float = 1.0, scale: bool = True, return_tensors: bool = False):
    """Batched psd2d over a (T, ny, nx) stack -> (T, ny, nx) float32."""
    torch = _ffi.require_gpu()
    t, _, src = _frames(stack, (3,))
    T, ny, nx = t.shape
    pl = _ffi.get_plan(ny, nx, _ffi.stack_chunk(ny, nx, int(T)))
    out = torch.empty((T, ny, nx), dtype=torch.float32, device=t.device)
    s = (dx * dy) / (float(nx) * float(ny)) if scale else 1.0
    _ffi.check(_ffi.lib().b4d_psd2d(pl.handle, D.ptr(t), int(T), D.ptr(out), float(s), _ffi.stream_ptr()))
    return _finish(out, True, return_tensors, np.float64 if src is np.float64 else np.float32)


def psd2d(image, *, x=None, y=None, dx: float = 1.0, dy: float = 1.0, scale: bool = True,
          return_tensors: bool = False):
    """Shifted 2-D PSD: |F|^2 * (dx*dy)/(nx*ny) when scale (reference: signal/fft.py:261-309)."""
    if not D.is_tensor(image):
        image = np.asarray(image)
    if image.ndim != 2:
        raise ValueError("image must be a 2D array.")
    ny, nx = image.shape
    sx, sy = _resolve_steps_2d(shape=(ny, nx), x=x, y=y, dx=dx, dy=dy)
    fx, fy = freq_axes2d(shape=(ny, nx), x=x, y=y, dx=dx, dy=dy)
    P = psd2d_stack(image[None], dx=sx, dy=sy, scale=scale, return_tensors=True)[0]
    return _finish(P, True, return_tensors, D.result_dtype(image)), fx, fy
