"""Frame-sharded translation tracking of a (T, H, W) stack across GPUs.

The tracking block of ``speckle_stack_stats`` (speckles.py:332-444) correlates every frame with templates cut from
frame 0 ("abs") and from the previous frame ("inc").  With the stack sharded frame-wise, one process per GPU, each rank
needs two frames it does not own: global frame 0 (one broadcast from rank 0) and the frame just before its shard (a
one-frame halo from the previous rank).  Everything else is local: no other collective on the data path.  The
per-frame results are small ((T_local, 3, 3) floats) and can be gathered with ``gather_series``.

Collectives go through torch.distributed: "nccl" (= RCCL over xGMI) on device tensors, "gloo" on CPU tensors in the
logic tests.
"""
from __future__ import annotations

import numpy as np

from .temporal import shard_bounds  # noqa: F401  (re-export: the same contiguous split)


def exchange_tracking_frames(local_stack, *, group=None):
    """(frame0, prev) for this rank: global frame 0 and the frame preceding this rank's first frame (rank 0: its own
    frame 0, as the reference does for t = 0).  local_stack: (T_local >= 1, H, W) torch tensor, CPU or device."""
    import torch
    import torch.distributed as dist

    first, last = local_stack[0], local_stack[-1]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return first.clone(), first.clone()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    frame0 = first.clone().contiguous()
    dist.broadcast(frame0, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    prev = first.clone().contiguous()
    ops = []
    if rank + 1 < world:
        dst = dist.get_global_rank(group, rank + 1) if group is not None else rank + 1
        ops.append(dist.P2POp(dist.isend, last.contiguous(), dst, group))
    if rank > 0:
        src = dist.get_global_rank(group, rank - 1) if group is not None else rank - 1
        ops.append(dist.P2POp(dist.irecv, prev, src, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return frame0, prev


def track_stack_sharded(local_stack, rois, *, method: str = "phase", backend: str = "internal", subpixel: bool = True,
                        eps: float = 1e-9, group=None, frame0=None, prev=None):
    """abs / inc shifts of this rank's frames on the ROI list `rois` [(y0, y1, x0, x1), ...].

    Returns {"dy_abs", "dx_abs", "dy_inc", "dx_inc"}: float32 arrays (T_local, len(rois)).  frame0 / prev override the
    exchange (single-process use and tests)."""
    from .. import _device as D
    from .. import _ffi
    from ..signal.tracking import phase_correlation_batch, template_matching_batch

    torch = _ffi.require_gpu()
    dev, _, _ = D.to_device_f32(local_stack, ndim=(3,))
    if frame0 is None or prev is None:
        f0, pv = exchange_tracking_frames(dev, group=group)
    else:
        f0, _, _ = D.to_device_f32(frame0, ndim=(2,))
        pv, _, _ = D.to_device_f32(prev, ndim=(2,))
    T, nr = int(dev.shape[0]), len(rois)
    # template sources: [frame0, prev, local frames...]; abs templates from source 0, inc templates of frame i from source i + 1
    src = torch.cat([f0[None], pv[None], dev], dim=0)
    tpl_frame = [0] * nr + [1 + i for i in range(T) for _ in range(nr)]
    tpl_roi = list(rois) + list(rois) * T
    pair_img = [i for i in range(T) for _ in range(nr)] * 2
    pair_tpl = [k for _ in range(T) for k in range(nr)] + [nr + nr * i + k for i in range(T) for k in range(nr)]
    if method.strip().lower() == "template":
        res = template_matching_batch(dev, src, tpl_frame, tpl_roi, pair_img, pair_tpl, backend=backend, subpixel=subpixel, eps=eps)
    else:
        if backend != "internal":
            raise ValueError("backend must be 'internal' for method='phase'.")
        res = phase_correlation_batch(dev, src, tpl_frame, tpl_roi, pair_img, pair_tpl, subpixel=subpixel, eps=eps)
    n = T * nr
    f32 = lambda a: a.reshape(T, nr).astype(np.float32)  # noqa: E731
    return {"dy_abs": f32(res[:n, 0]), "dx_abs": f32(res[:n, 1]), "dy_inc": f32(res[n:, 0]), "dx_inc": f32(res[n:, 1])}


def gather_series(local: np.ndarray, *, group=None) -> np.ndarray:
    """Concatenate per-rank (T_local, ...) arrays along axis 0 on every rank (shards may differ in length)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, local, group=group)
    return np.concatenate(parts, axis=0)
