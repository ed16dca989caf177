"""Per-pixel temporal statistics of a (T, H, W) stack, sharded frame-wise across GPUs.

BASELINE.json config 4 / SURVEY.md §8 rows a23 + 8(e).  The reference only has the temporal
mean (``data.mean(axis=0)``, io/rw.py:129-132); variance and contrast follow the package's
ddof=0 convention: mean_t = sum(x)/T, var_t = sum(x^2)/T - mean_t^2, contrast_t = sqrt(var_t)/mean_t.

Each rank streams its own frames once (float64 accumulators, b4d_temporal_accumulate*), then ONE
all-reduce(sum) crosses xGMI (RCCL through torch.distributed, backend "nccl"; "gloo" on CPU tensors for
the logic tests): the frame count rides in the SAME float64 buffer as the sums,

    buf = [count, 0, | sum_x rows of chunk 0 | sum_xx rows of chunk 0 | chunk 1 ... ]

and ``b4d_temporal_finalize_dev`` reads it from device memory, so the host never waits for the collective.
``overlap_chunks = k > 1`` splits the image rows into k blocks: block c is accumulated over all local frames,
then its slice of ``buf`` is all-reduced on a side stream while block c + 1 is still being accumulated
(the collective is 4-25 % of a cfg4 step at 8 GPUs) -- k collectives of 1/k of the payload each,
the first one carrying the count.

``collective="reduce_scatter"`` is SURVEY.md §8e's all-links alternative: the image rows are cut into one equal slice per rank,
ONE reduce-scatter leaves every rank with the summed [count, sum_x, sum_xx] of ITS slice (1/N of the all-reduce's payload per
link direction), each rank finalises its slice, and ONE all-gather of the three float32 maps (12 instead of 16 bytes per pixel)
rebuilds them everywhere.  Same sums, same finalisation kernel: the maps are bit-identical to the all-reduce route's.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _ffi


def _dist_group(group):
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():   # a 1-rank group still runs the (trivial) collective: same code path
        return dist
    return None


class TemporalSums:
    """float64 accumulators of one rank in the packed, chunk-major layout described above."""

    HEAD = 2   # [count, pad]: keeps every block 16-byte aligned

    def __init__(self, H: int, W: int, device, chunks: int = 1):
        import torch

        self.H, self.W = int(H), int(W)
        chunks = max(1, min(int(chunks), self.H))
        edges = [round(i * self.H / chunks) for i in range(chunks + 1)]
        self.rows = [(edges[i], edges[i + 1]) for i in range(chunks) if edges[i + 1] > edges[i]]
        self.buf = torch.zeros(self.HEAD + 2 * self.H * self.W, dtype=torch.float64, device=device)
        self.offsets = [self.HEAD + 2 * r0 * self.W for r0, _ in self.rows]

    def block(self, c: int):
        """(sum_x, sum_xx) views of row block c, each (rows, W)."""
        r0, r1 = self.rows[c]
        n = (r1 - r0) * self.W
        o = self.offsets[c]
        return self.buf[o:o + n].view(r1 - r0, self.W), self.buf[o + n:o + 2 * n].view(r1 - r0, self.W)

    def slice_for_reduce(self, c: int):
        """Contiguous slice of `buf` holding block c (block 0 also carries the count)."""
        r0, r1 = self.rows[c]
        o = self.offsets[c]
        return self.buf[(0 if c == 0 else o):o + 2 * (r1 - r0) * self.W]

    def add_count(self, n: int):
        self.buf[0] += float(n)

    def accumulate(self, frames, c: int | None = None, stream=None):
        """sums += over `frames` (n, H, W) float32 device tensor; c = one row block, None = all of them."""
        n, H, W = (int(v) for v in frames.shape)
        assert (H, W) == (self.H, self.W) and frames.is_contiguous()
        lib = _ffi.lib()
        st = _ffi.stream_ptr() if stream is None else C.c_void_p(stream.cuda_stream)
        for cc in (range(len(self.rows)) if c is None else (c,)):
            r0, r1 = self.rows[cc]
            sx, sxx = self.block(cc)
            _ffi.check(lib.b4d_temporal_accumulate_range(C.c_void_p(frames.data_ptr()), n, H * W, r0 * W, (r1 - r0) * W,
                                                         C.c_void_p(sx.data_ptr()), C.c_void_p(sxx.data_ptr()), st))

    def finalize(self, c: int, mean, var, con):
        """Row block c of the float32 maps from the (all-reduced) sums; the count comes from buf[0] on the device."""
        r0, r1 = self.rows[c]
        sx, sxx = self.block(c)
        _ffi.check(_ffi.lib().b4d_temporal_finalize_dev(
            C.c_void_p(sx.data_ptr()), C.c_void_p(sxx.data_ptr()), C.c_void_p(self.buf.data_ptr()), (r1 - r0) * self.W,
            C.c_void_p(mean[r0:r1].data_ptr()), C.c_void_p(var[r0:r1].data_ptr()), C.c_void_p(con[r0:r1].data_ptr()),
            _ffi.stream_ptr()))


def scatter_layout(H: int, W: int, world: int):
    """Row slices of the reduce-scatter route: `world` equal slices of `rows` image rows (the last ones padded), each packed as
    [count, pad, sum_x (rows W), sum_xx (rows W)] so that every rank's share of the reduce-scatter carries the frame count."""
    rows = -(-int(H) // int(world))
    return rows, TemporalSums.HEAD + 2 * rows * int(W)


def _reduce_scatter(dist, out, inp, group):
    """out = this rank's slice of sum over ranks of inp.  RCCL ("nccl") has the collective; gloo (the CPU logic tests) does not in
    every build: there the same result comes from an all-reduce and a slice."""
    try:
        dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)
    except (RuntimeError, NotImplementedError):
        if dist.get_backend(group) != "gloo":
            raise
        dist.all_reduce(inp, op=dist.ReduceOp.SUM, group=group)
        r = dist.get_rank(group)
        out.copy_(inp[r * out.numel():(r + 1) * out.numel()])


def _temporal_stats_scatter(torch, dist, group, t, chunk, timings):
    """The reduce-scatter + all-gather route of temporal_stats (module docstring)."""
    T, H, W = (int(v) for v in t.shape)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rows, per = scatter_layout(H, W, world)
    buf = torch.zeros(world * per, dtype=torch.float64, device=t.device)
    buf.view(world, per)[:, 0] = float(T)
    lib, st = _ffi.lib(), _ffi.stream_ptr()
    for r in range(world):
        r0, r1 = r * rows, min((r + 1) * rows, H)
        if r1 <= r0:
            continue
        o = r * per + TemporalSums.HEAD
        for a in range(0, T, chunk):
            fr = t[a:a + chunk]
            _ffi.check(lib.b4d_temporal_accumulate_range(C.c_void_p(fr.data_ptr()), int(fr.shape[0]), H * W, r0 * W, (r1 - r0) * W,
                                                         C.c_void_p(buf[o:].data_ptr()), C.c_void_p(buf[o + rows * W:].data_ptr()), st))
    mine = torch.empty(per, dtype=torch.float64, device=t.device)
    ev = None
    if timings is not None:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
    _reduce_scatter(dist, mine, buf, group)
    if ev:
        ev[1].record()
    maps = torch.zeros((3, rows * W), dtype=torch.float32, device=t.device)
    sx, sxx = mine[TemporalSums.HEAD:TemporalSums.HEAD + rows * W], mine[TemporalSums.HEAD + rows * W:]
    _ffi.check(lib.b4d_temporal_finalize_dev(C.c_void_p(sx.data_ptr()), C.c_void_p(sxx.data_ptr()), C.c_void_p(mine.data_ptr()), rows * W,
                                             C.c_void_p(maps[0].data_ptr()), C.c_void_p(maps[1].data_ptr()), C.c_void_p(maps[2].data_ptr()), st))
    gathered = torch.empty((world, 3, rows * W), dtype=torch.float32, device=t.device)
    if ev:
        ev[2].record()
    dist.all_gather_into_tensor(gathered.view(-1), maps.view(-1), group=group)
    if ev:
        ev[3].record()
        ev[3].synchronize()
        timings["reduce_scatter_ms"] = float(ev[0].elapsed_time(ev[1]))
        timings["all_gather_ms"] = float(ev[2].elapsed_time(ev[3]))
    full = gathered.permute(1, 0, 2).reshape(3, world * rows, W)[:, :H]
    return full[0].contiguous(), full[1].contiguous(), full[2].contiguous()


def temporal_stats(local_stack, *, group=None, chunk: int = 1024, overlap_chunks: int = 1, return_tensors: bool = False,
                   timings: dict | None = None, collective: str = "all_reduce"):
    """mean / variance / contrast maps over ALL frames of all ranks.

    local_stack: this rank's frames (T_local, H, W), NumPy or ROCm tensor (float32 used; any H, W).
    overlap_chunks: row blocks whose all-reduce overlaps the accumulation of the following blocks (1 = one collective).
    collective: "all_reduce" (one all-reduce of the packed sums) or "reduce_scatter" (reduce-scatter of row slices, local
        finalisation, all-gather of the float32 maps: module docstring); without a process group both are the local computation.
    timings: optional dict receiving {"allreduce_ms": ...} or {"reduce_scatter_ms", "all_gather_ms"} (HIP events around the
        collectives; forces a device sync).
    Returns (mean, var, contrast) float32 (H, W)."""
    torch = _ffi.require_gpu()
    from .. import _device as D

    if collective not in ("all_reduce", "reduce_scatter"):
        raise ValueError("collective must be 'all_reduce' or 'reduce_scatter'.")
    t, _, _ = D.to_device_f32(local_stack, ndim=(3,))
    T, H, W = (int(v) for v in t.shape)
    dist = _dist_group(group)
    if dist is not None and collective == "reduce_scatter":
        mean, var, con = _temporal_stats_scatter(torch, dist, group, t, chunk, timings)
        if return_tensors:
            return mean, var, con
        return mean.cpu().numpy(), var.cpu().numpy(), con.cpu().numpy()
    acc = TemporalSums(H, W, t.device, overlap_chunks if dist is not None else 1)
    acc.add_count(T)
    mean = torch.empty((H, W), dtype=torch.float32, device=t.device)
    var, con = torch.empty_like(mean), torch.empty_like(mean)
    main = torch.cuda.current_stream()
    nblk = len(acc.rows)
    side = torch.cuda.Stream() if dist is not None and nblk > 1 else None
    ev0 = ev1 = None
    for c in range(nblk):
        for a in range(0, T, chunk):
            acc.accumulate(t[a:a + chunk], c)
        if dist is None:
            acc.finalize(c, mean, var, con)
            continue
        if side is None:      # ONE all-reduce of count + sums, on the caller's stream
            if timings is not None:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(main)
            dist.all_reduce(acc.slice_for_reduce(c), op=dist.ReduceOp.SUM, group=group)
            if ev1 is not None:
                ev1.record(main)
            acc.finalize(c, mean, var, con)
        else:                 # block c's collective + finalize on the side stream, block c + 1 accumulates meanwhile
            side.wait_stream(main)
            with torch.cuda.stream(side):
                dist.all_reduce(acc.slice_for_reduce(c), op=dist.ReduceOp.SUM, group=group)
                acc.finalize(c, mean, var, con)
    if side is not None:
        main.wait_stream(side)
    if timings is not None and ev1 is not None:
        ev1.synchronize()
        timings["allreduce_ms"] = float(ev0.elapsed_time(ev1))
    if return_tensors:
        return mean, var, con
    return mean.cpu().numpy(), var.cpu().numpy(), con.cpu().numpy()


def shard_bounds(total_frames: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous frame range [t0, t1) of `rank`: remainders go to the first ranks."""
    base, rem = divmod(int(total_frames), int(world_size))
    t0 = rank * base + min(rank, rem)
    return t0, t0 + base + (1 if rank < rem else 0)


def scatter_reduce_sums_cpu(sum_x: np.ndarray, sum_xx: np.ndarray, count: int, finalize, group=None):
    """The reduce-scatter + all-gather route on host arrays (gloo) for the multi-process CPU tests: same slice layout, same two
    collectives as _temporal_stats_scatter; `finalize(sum_x, sum_xx, count)` -> (mean, var, con) stands in for the device kernel."""
    import torch

    dist = _dist_group(group)
    H, W = sum_x.shape
    world = dist.get_world_size(group) if dist is not None else 1
    rank = dist.get_rank(group) if dist is not None else 0
    rows, per = scatter_layout(H, W, world)
    buf = torch.zeros((world, per), dtype=torch.float64)
    buf[:, 0] = float(count)
    for r in range(world):
        r0, r1 = r * rows, min((r + 1) * rows, H)
        if r1 > r0:
            n = (r1 - r0) * W
            buf[r, TemporalSums.HEAD:TemporalSums.HEAD + n] = torch.from_numpy(np.ascontiguousarray(sum_x[r0:r1], dtype=np.float64).ravel())
            buf[r, TemporalSums.HEAD + rows * W:TemporalSums.HEAD + rows * W + n] = torch.from_numpy(
                np.ascontiguousarray(sum_xx[r0:r1], dtype=np.float64).ravel())
    mine = torch.empty(per, dtype=torch.float64)
    if dist is not None:
        _reduce_scatter(dist, mine, buf.view(-1), group)
    else:
        mine.copy_(buf[0])
    m = mine.numpy()
    with np.errstate(all="ignore"):
        maps = finalize(m[TemporalSums.HEAD:TemporalSums.HEAD + rows * W].reshape(rows, W),
                        m[TemporalSums.HEAD + rows * W:].reshape(rows, W), float(m[0]))
    local = torch.from_numpy(np.stack([np.asarray(x, dtype=np.float32) for x in maps]).reshape(-1).copy())
    gathered = torch.empty(world * local.numel(), dtype=torch.float32)
    if dist is not None:
        dist.all_gather_into_tensor(gathered, local, group=group)
    else:
        gathered.copy_(local)
    full = gathered.view(world, 3, rows, W).permute(1, 0, 2, 3).reshape(3, world * rows, W)[:, :H].numpy()
    return full[0].copy(), full[1].copy(), full[2].copy(), float(m[0])


def reduce_sums_cpu(sum_x: np.ndarray, sum_xx: np.ndarray, count: int, group=None):
    """The same packed collective on host float64 arrays (gloo) -- used by the multi-process CPU tests of the
    sharding / reduction logic: ONE all-reduce of [count, pad, sum_x, sum_xx]; returns (sum_x, sum_xx, count) over all ranks."""
    import torch

    n = sum_x.size
    buf = torch.zeros(TemporalSums.HEAD + 2 * n, dtype=torch.float64)
    buf[0] = float(count)
    buf[TemporalSums.HEAD:TemporalSums.HEAD + n] = torch.from_numpy(np.ascontiguousarray(sum_x, dtype=np.float64).ravel())
    buf[TemporalSums.HEAD + n:] = torch.from_numpy(np.ascontiguousarray(sum_xx, dtype=np.float64).ravel())
    dist = _dist_group(group)
    if dist is not None:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    out = buf.numpy()
    return (out[TemporalSums.HEAD:TemporalSums.HEAD + n].reshape(sum_x.shape).copy(),
            out[TemporalSums.HEAD + n:].reshape(sum_xx.shape).copy(), float(out[0]))
