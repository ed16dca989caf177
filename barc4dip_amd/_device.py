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
    if arr.nbytes >= _UPLOAD_MIN_BYTES and arr.flags.c_contiguous and arr.dtype.name in _UPLOAD_CODES and arr.dtype != np.float32:
        return _upload_staged(arr), False, src_dtype
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to("cuda", non_blocking=False)
    return t, False, src_dtype


# ---- large host arrays that are not float32 (the host -> HBM half): a host-side astype of detector words runs
# at 1-4 GB/s on one core.  Integer and float64 arrays of 32 MiB and more go up in their NATIVE dtype (uint16 detector words stay
# 2 bytes on the bus) through two page-locked 32-MiB blocks filled by a few host threads (NumPy releases the GIL while copying)
# while the previous block is on the bus, and are converted to float32 on the device (b4d_to_f32: the same round-to-nearest
# conversion as ndarray.astype).  8 x 2048^2 frames (tools/dev_upload.py): uint16 16.2 -> 2.4 ms, float64 21.4 -> 7.4 ms; float32
# arrays keep the plain copy (2.7 ms against 3.3 staged: the runtime's own staging is as fast).  Everything stays on the caller's
# stream.
_UPLOAD_MIN_BYTES = 32 << 20
_UPLOAD_BLOCK = 32 << 20
_UPLOAD_CODES = {"uint8": 0, "uint16": 1, "int16": 2, "int32": 3, "uint32": 4, "float32": 5, "float64": 6}
_upload_pool = None


def _upload_staged(arr: np.ndarray):
    global _upload_pool
    torch = _ffi.require_gpu()
    from concurrent.futures import ThreadPoolExecutor

    if _upload_pool is None:
        _upload_pool = ThreadPoolExecutor(max_workers=4, thread_name_prefix="b4d-upload")
    lib = _ffi.lib()
    code, item = _UPLOAD_CODES[arr.dtype.name], arr.dtype.itemsize
    flat = arr.reshape(-1)
    n = int(flat.size)
    out = torch.empty(arr.shape, dtype=torch.float32, device="cuda")
    oflat = out.view(-1)
    per = _UPLOAD_BLOCK // item
    pinned = [torch.empty(per * item, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    raw = None if code == 5 else [torch.empty(per * item, dtype=torch.uint8, device="cuda") for _ in range(2)]
    sent = [None, None]
    stream = torch.cuda.current_stream()
    for k, a in enumerate(range(0, n, per)):
        b, slot = min(n, a + per), k & 1
        if sent[slot] is not None:
            sent[slot].synchronize()        # the block's previous copy has left the page-locked buffer
        host = pinned[slot].numpy()[:(b - a) * item].view(arr.dtype)
        q = max(1, -(-(b - a) // 4))
        list(_upload_pool.map(lambda lo: np.copyto(host[lo:lo + q], flat[a + lo:min(b, a + lo + q)]), range(0, b - a, q)))
        if code == 5:
            oflat[a:b].copy_(pinned[slot][:(b - a) * item].view(torch.float32), non_blocking=True)
        else:
            raw[slot][:(b - a) * item].copy_(pinned[slot][:(b - a) * item], non_blocking=True)
            _ffi.check(lib.b4d_to_f32(C.c_void_p(raw[slot].data_ptr()), code, b - a, C.c_void_p(oflat[a:b].data_ptr()),
                                      C.c_void_p(stream.cuda_stream)))
        ev = torch.cuda.Event()
        ev.record(stream)
        sent[slot] = ev
    for ev in sent:
        if ev is not None:
            ev.synchronize()                # the page-locked blocks go back to torch's host allocator only once they are idle
    return out


def ptr(t) -> C.c_void_p:
    return C.c_void_p(t.data_ptr())


_PINNED_MIN_BYTES = 1 << 20      # below this a plain .cpu() is as fast
_PINNED_MAX_BYTES = 1 << 30      # above this the result is not page-locked as a whole: staged through two blocks (_download_staged)


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
    if t.is_cuda and nbytes > _PINNED_MAX_BYTES and t.is_contiguous():
        return _download_staged(t, want)
    if not t.is_cuda or nbytes < _PINNED_MIN_BYTES or nbytes > _PINNED_MAX_BYTES:
        out = t.cpu().numpy()
        return out.astype(dtype, copy=False) if dtype is not None else out
    host = torch.empty(tuple(t.shape), dtype=want, pin_memory=True)
    host.copy_(t if want == t.dtype else t.to(want), non_blocking=True)
    torch.cuda.current_stream().synchronize()
    return host.numpy()


def _download_staged(t, want):
    """Results beyond 256 MB (the (T, N, N) float64 autocorrelation stack of speckle_stack_stats ...): converted on the device block
    by block, copied into two page-locked 32-MiB blocks and from there into the (pageable) result by a few host threads while the
    next block is on the bus -- instead of one pageable copy + a single-threaded astype of the whole array."""
    global _upload_pool
    torch = _ffi.require_gpu()
    from concurrent.futures import ThreadPoolExecutor

    if _upload_pool is None:
        _upload_pool = ThreadPoolExecutor(max_workers=4, thread_name_prefix="b4d-upload")
    item = torch.empty(0, dtype=want).element_size()
    out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=want).numpy().dtype)
    oflat, tflat = out.reshape(-1), t.reshape(-1)
    n, per = int(tflat.numel()), _UPLOAD_BLOCK // item
    pinned = [torch.empty(per, dtype=want, pin_memory=True) for _ in range(2)]
    ready = [None, None]
    spans = [(a, min(n, a + per)) for a in range(0, n, per)]
    stream = torch.cuda.current_stream()

    def drain(k):
        a, b = spans[k]
        ready[k & 1].synchronize()
        host = pinned[k & 1].numpy()[:b - a]
        q = max(1, -(-(b - a) // 4))
        list(_upload_pool.map(lambda lo: np.copyto(oflat[a + lo:min(b, a + lo + q)], host[lo:lo + q]), range(0, b - a, q)))

    for k, (a, b) in enumerate(spans):
        if k >= 2:
            drain(k - 2)                    # the block this one will overwrite has been copied out
        src = tflat[a:b]
        pinned[k & 1][:b - a].copy_(src if want == t.dtype else src.to(want), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        ready[k & 1] = ev
    for k in range(max(0, len(spans) - 2), len(spans)):
        drain(k)
    return out


def result_dtype(a):
    """Real dtype the reference's NumPy pipeline would return for input `a`: float32 for float16 / float32 input,
    float64 for everything else (integers, bool, float64) -- numpy.fft promotes those to double precision."""
    name = str(getattr(a, "dtype", "float64")).replace("torch.", "")
    return np.float32 if name in ("float32", "float16", "bfloat16") else np.float64
