"""Host<->device plumbing: accept NumPy arrays or ROCm torch tensors, hand out device pointers.

PyTorch is used only for device memory, streams and (elsewhere) torch.distributed; all compute
goes through the C ABI in _ffi.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi


def is_tensor(a) -> bool:
    try:
        import torch
    except Exception:  # pragma: no cover
        return False
    return isinstance(a, torch.Tensor)


def to_device_f32(a, *, ndim: tuple[int, ...]):
    """Return (contiguous float32 CUDA tensor, was_tensor, numpy_dtype_of_input)."""
    torch = _ffi.require_gpu()
    if is_tensor(a):
        if a.is_complex():
            raise NotImplementedError("complex input is not supported by the HIP path (real frames only).")
        if a.ndim not in ndim:
            raise ValueError(f"expected ndim in {ndim}, got {a.ndim}")
        src_dtype = np.float32 if a.dtype in (torch.float32, torch.float16, torch.bfloat16) else np.float64
        return a.to(device="cuda", dtype=torch.float32).contiguous(), True, src_dtype
    arr = np.asarray(a)
    if np.iscomplexobj(arr):
        raise NotImplementedError("complex input is not supported by the HIP path (real frames only).")
    if arr.ndim not in ndim:
        raise ValueError(f"expected ndim in {ndim}, got {arr.ndim}")
    # NumPy's FFT promotes everything except float16 / float32 to double precision: outputs follow that dtype
    src_dtype = np.float32 if arr.dtype in (np.float32, np.float16) else np.float64
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to("cuda", non_blocking=False)
    return t, False, src_dtype


def ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


_PINNED_MIN_BYTES = 1 << 20      # below this a plain .cpu() is as fast
_PINNED_MAX_BYTES = 256 << 20    # above this the result is not worth page-locking: pageable copy


def to_host(t, dtype=None) -> np.ndarray:
    """Device tensor -> NumPy array of `dtype` (default: the tensor's).  Mid-sized results (the (N, N) autocorrelation map that
    `grain` returns as float64, PSD / spectrum frames ...) are converted ON THE DEVICE and copied into a page-locked host block of
    their own, which the returned array keeps alive (torch's caching host allocator hands the block out again once the array is
    gone): a 2048^2 float32 map -> float64 array took 4.1 ms (pageable copy) + 3.3 ms (single-threaded astype) and was most of
    speckle_stats' wall clock; this way it is under a millisecond."""
    torch = _ffi.require_gpu()
    t = t.detach()
    want = t.dtype if dtype is None else torch.from_numpy(np.empty(0, dtype=dtype)).dtype
    nbytes = t.numel() * torch.empty(0, dtype=want).element_size()
    if not t.is_cuda or nbytes < _PINNED_MIN_BYTES or nbytes > _PINNED_MAX_BYTES:
        out = t.cpu().numpy()
        return out.astype(dtype, copy=False) if dtype is not None else out
    host = torch.empty(tuple(t.shape), dtype=want, pin_memory=True)
    host.copy_(t if want == t.dtype else t.to(want), non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.numpy()


def result_dtype(a):
    """Real dtype the reference's NumPy pipeline would return for input `a`: float32 for float16 / float32 input,
    float64 for everything else (integers, bool, float64) -- numpy.fft promotes those to double precision."""
    name = str(getattr(a, "dtype", "float64")).replace("torch.", "")
    return np.float32 if name in ("float32", "float16", "bfloat16") else np.float64
