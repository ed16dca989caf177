"""Per-pixel temporal statistics of a (T, H, W) stack, sharded frame-wise across GPUs.

BASELINE.json config 4 / SURVEY.md §8 rows a23 + 8(e).  The reference only has the temporal
mean (``data.mean(axis=0)``, io/rw.py:129-132); variance and contrast follow the package's
ddof=0 convention: mean_t = sum(x)/T, var_t = sum(x^2)/T - mean_t^2, contrast_t = sqrt(var_t)/mean_t.

Each rank streams its own frames once (float64 accumulators, b4d_temporal_accumulate), then ONE
all-reduce(sum) of the stacked (2, H, W) float64 sums + frame count crosses xGMI (RCCL through
torch.distributed, backend "nccl"; "gloo" on CPU tensors for the logic tests).
"""
from __future__ import annotations

import numpy as np

from .. import _ffi
from . import kernels as K


def _reduce(sums, count, group=None):
    """All-reduce the (2, H, W) sums and the frame count.  No-op without an initialised process group."""
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        cnt = torch.tensor([float(count)], dtype=torch.float64, device=sums.device)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
        count = float(cnt.item())
    return sums, count


def temporal_stats(local_stack, *, group=None, chunk: int = 1024, return_tensors: bool = False):
    """mean / variance / contrast maps over ALL frames of all ranks.

    local_stack: this rank's frames (T_local, H, W), NumPy or ROCm tensor (float32 used).
    Returns (mean, var, contrast) float32 (H, W)."""
    torch = _ffi.require_gpu()
    from .. import _device as D

    t, _, _ = D.to_device_f32(local_stack, ndim=(3,))
    T, H, W = (int(v) for v in t.shape)
    sums = torch.zeros((2, H, W), dtype=torch.float64, device=t.device)
    for a in range(0, T, chunk):
        K.temporal_accumulate(t[a:a + chunk], sums[0], sums[1])
    sums, count = _reduce(sums, T, group)
    mean, var, con = K.temporal_finalize(sums[0], sums[1], count)
    if return_tensors:
        return mean, var, con
    return mean.cpu().numpy(), var.cpu().numpy(), con.cpu().numpy()


def shard_bounds(total_frames: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous frame range [t0, t1) of `rank`: remainders go to the first ranks."""
    base, rem = divmod(int(total_frames), int(world_size))
    t0 = rank * base + min(rank, rem)
    return t0, t0 + base + (1 if rank < rem else 0)


def reduce_sums_cpu(sum_x: np.ndarray, sum_xx: np.ndarray, count: int, group=None):
    """The same collective on host float64 arrays (gloo) -- used by the multi-process CPU tests of
    the sharding/reduction logic; returns (sum_x, sum_xx, count) over all ranks."""
    import torch

    s = torch.from_numpy(np.stack([sum_x, sum_xx]).astype(np.float64))
    s, c = _reduce(s, count, group)
    return s[0].numpy(), s[1].numpy(), c
