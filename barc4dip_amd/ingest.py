"""Host -> HBM streaming of frame stacks that do not fit (or do not live) in device memory.

The reference reads whole stacks into host arrays (io/rw.py, io/h5.py); h5py is not available in this image, so the
HDF5 side is not built.  What is here is the device half of that row: any (T, H, W) array-like that supports slicing
(``numpy.ndarray``, ``numpy.memmap`` of a .npy file, an ``h5py.Dataset`` where h5py exists) is cut into chunks, each
chunk is copied in its NATIVE dtype (uint16 detector words stay 2 bytes on the bus) into one of two PINNED staging
buffers by a few host threads, sent to the device on a side stream and converted to float32 there (b4d_to_f32) while
the previous chunk is being processed, so disk / PCIe overlap the kernels.

PyTorch provides pinned memory, streams and events (plumbing); the arithmetic stays in libb4d.
"""
from __future__ import annotations

import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _ffi

_DTYPE_CODES = {"uint8": 0, "uint16": 1, "int16": 2, "int32": 3, "uint32": 4, "float32": 5, "float64": 6}


def iter_device_chunks(source, chunk_frames: int = 16):
    """Yield float32 device tensors (n <= chunk_frames, H, W) covering ``source`` in order.

    Each yielded tensor is valid until the NEXT iteration step: its storage is one of two rotating device buffers.
    Work must be queued on the current torch stream (as every barc4dip_amd entry point does)."""
    torch = _ffi.require_gpu()
    if len(source.shape) != 3:
        raise ValueError(f"source must have shape (T, H, W); got {tuple(source.shape)}")
    T, H, W = (int(v) for v in source.shape)
    chunk = max(1, min(int(chunk_frames), T))
    name = np.dtype(source.dtype).name
    code = _DTYPE_CODES.get(name)
    raw_dtype = np.dtype(source.dtype) if code is not None else np.dtype(np.float32)   # other dtypes: converted on the host
    code = code if code is not None else _DTYPE_CODES["float32"]
    item = raw_dtype.itemsize
    copy_stream = torch.cuda.Stream()
    nbytes = chunk * H * W * item
    pinned = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
    raw_dev = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
    device = [torch.empty((chunk, H, W), dtype=torch.float32, device="cuda") for _ in range(2)]
    copied = [None, None]      # H2D + conversion of slot i finished (recorded on the copy stream)
    consumed = [None, None]    # consumer's kernels on slot i were queued before this event (current stream)
    pool = ThreadPoolExecutor(max_workers=4)
    lib = _ffi.lib()

    def stage(slot: int, start: int) -> int:
        n = min(chunk, T - start)
        if copied[slot] is not None:
            copied[slot].synchronize()                      # the previous H2D out of this pinned buffer is done
        host = pinned[slot].numpy()[:n * H * W * item].view(raw_dtype).reshape(n, H, W)
        parts = [(a, min(n, a + max(1, (n + 3) // 4))) for a in range(0, n, max(1, (n + 3) // 4))]
        # NumPy releases the GIL while copying: a few threads fill the pinned buffer at memory speed
        list(pool.map(lambda ab: np.copyto(host[ab[0]:ab[1]], source[start + ab[0]:start + ab[1]], casting="unsafe"), parts))
        with torch.cuda.stream(copy_stream):
            if consumed[slot] is not None:
                copy_stream.wait_event(consumed[slot])      # do not overwrite a chunk that is still being read
            raw_dev[slot][:n * H * W * item].copy_(pinned[slot][:n * H * W * item], non_blocking=True)
            _ffi.check(lib.b4d_to_f32(C.c_void_p(raw_dev[slot].data_ptr()), code, n * H * W, C.c_void_p(device[slot].data_ptr()),
                                      C.c_void_p(copy_stream.cuda_stream)))
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        copied[slot] = ev
        return n

    sizes = {0: stage(0, 0)}
    start, slot = 0, 0
    while start < T:
        n = sizes[slot]
        nxt = start + n
        if nxt < T:
            sizes[1 - slot] = stage(1 - slot, nxt)          # next chunk in flight while this one is processed
        torch.cuda.current_stream().wait_event(copied[slot])
        yield device[slot][:n]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        consumed[slot] = ev
        start, slot = nxt, 1 - slot
    pool.shutdown(wait=False)


def temporal_stats_streamed(source, *, chunk_frames: int = 16, group=None, return_tensors: bool = False):
    """metrics.temporal_stats over a host / memory-mapped stack streamed through iter_device_chunks: per-pixel
    mean, variance and contrast maps (float32) of ALL frames (and all ranks when a process group is given)."""
    torch = _ffi.require_gpu()
    from .metrics.temporal import TemporalSums, _dist_group

    T, H, W = (int(v) for v in source.shape)
    acc = TemporalSums(H, W, torch.device("cuda", torch.cuda.current_device()))
    acc.add_count(T)
    for dev in iter_device_chunks(source, chunk_frames):
        acc.accumulate(dev)
    dist = _dist_group(group)
    if dist is not None:     # ONE all-reduce: the count rides in the sums' buffer
        dist.all_reduce(acc.slice_for_reduce(0), op=dist.ReduceOp.SUM, group=group)
    mean = torch.empty((H, W), dtype=torch.float32, device=acc.buf.device)
    var, con = torch.empty_like(mean), torch.empty_like(mean)
    acc.finalize(0, mean, var, con)
    if return_tensors:
        return mean, var, con
    return mean.cpu().numpy(), var.cpu().numpy(), con.cpu().numpy()
