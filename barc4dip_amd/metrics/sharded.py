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


TRACK_WORKSPACE_BYTES = 4 << 30   # default bound on the half spectra one tracking call keeps resident


def track_abs_inc(dev, frame0, prev, rois, *, method: str = "phase", backend: str = "internal", subpixel: bool = True,
                  eps: float = 1e-9, workspace_bytes: int | None = None):
    """abs / inc shifts of the device frames `dev` (T, H, W) on the ROI list `rois`, in blocks of frames.

    abs templates are cut from `frame0`, the inc template of frame i from frame i - 1 (`prev` for i = 0).  The C entry
    points keep one half spectrum per image AND per template resident (b4d_phase_correlation: (nimg + ntpl) * 4 H W
    bytes), i.e. (1 + nroi) spectra per frame: a whole stack in one call would need ~43 GB for 256 frames of 2048^2.
    Frames therefore go through in blocks sized to `workspace_bytes` (default 4 GiB); a block re-transforms only the
    `nroi` abs templates, and its template sources are a (block + 1)-frame copy, never a copy of the stack.
    Returns float64 (T * nroi, 4) rows {dy, dx, peak, snr} for abs and for inc (frame-major, ROI-minor)."""
    import torch

    from ..signal.tracking import phase_correlation_batch, template_matching_batch

    T, H, W = (int(v) for v in dev.shape)
    nr = len(rois)
    budget = int(workspace_bytes or TRACK_WORKSPACE_BYTES)
    use_template = method.strip().lower() == "template"
    # per frame: 1 image + nroi inc-template half spectra (4 H W bytes each); the NCC matcher adds two float64 window-sum
    # tables per image (16 H W bytes = four more)
    blk = max(1, min(T, (budget // (4 * H * W) - nr) // (1 + nr + (4 if use_template else 0))))
    if not use_template and backend not in ("internal", "skimage"):
        raise ValueError("backend must be 'internal' or 'skimage' for method='phase'.")
    if not use_template and backend == "skimage":
        # up-sampled phase cross-correlation (signal.tracking._phase_correlation_upsampled): device transforms, host refinement,
        # one pair at a time -- the convenience route of that back-end, not a throughput path
        from .. import _device as D
        from ..signal.tracking import phase_correlation

        f0h, prevh = D.to_host(frame0), D.to_host(prev)
        ra, ri = np.empty((T * nr, 4)), np.empty((T * nr, 4))
        for t in range(T):
            cur = D.to_host(dev[t])
            for k, (y0, y1, x0, x1) in enumerate(rois):
                sl = (slice(int(y0), int(y1)), slice(int(x0), int(x1)))
                ra[t * nr + k] = phase_correlation(f0h[sl], cur, slices_yx=sl, backend="skimage", subpixel=subpixel, eps=eps)
                ri[t * nr + k] = phase_correlation(prevh[sl], cur, slices_yx=sl, backend="skimage", subpixel=subpixel, eps=eps)
            prevh = cur
        return ra, ri
    res_abs, res_inc = [], []
    for a in range(0, T, blk):
        b = min(T, a + blk)
        n = b - a
        # template sources of the block: [frame0, frame a - 1, frames a .. b - 2]
        src = torch.cat([frame0[None], prev[None] if a == 0 else dev[a - 1:a], dev[a:b - 1]], dim=0)
        tpl_frame = [0] * nr + [1 + i for i in range(n) for _ in range(nr)]
        tpl_roi = list(rois) + list(rois) * n
        pair_img = [i for i in range(n) for _ in range(nr)] * 2
        pair_tpl = [k for _ in range(n) for k in range(nr)] + [nr + nr * i + k for i in range(n) for k in range(nr)]
        if use_template:
            res = template_matching_batch(dev[a:b], src, tpl_frame, tpl_roi, pair_img, pair_tpl, backend=backend,
                                          subpixel=subpixel, eps=eps)
        else:
            res = phase_correlation_batch(dev[a:b], src, tpl_frame, tpl_roi, pair_img, pair_tpl, subpixel=subpixel, eps=eps)
        res_abs.append(res[:n * nr])
        res_inc.append(res[n * nr:])
    return np.concatenate(res_abs, axis=0), np.concatenate(res_inc, axis=0)


def track_stack_sharded(local_stack, rois, *, method: str = "phase", backend: str = "internal", subpixel: bool = True,
                        eps: float = 1e-9, group=None, frame0=None, prev=None, workspace_bytes: int | None = None):
    """abs / inc shifts of this rank's frames on the ROI list `rois` [(y0, y1, x0, x1), ...].

    Returns {"dy_abs", "dx_abs", "dy_inc", "dx_inc"}: float32 arrays (T_local, len(rois)).  frame0 / prev override the
    exchange (single-process use and tests)."""
    from .. import _device as D
    from .. import _ffi

    _ffi.require_gpu()
    dev, _, _ = D.to_device_f32(local_stack, ndim=(3,))
    if frame0 is None or prev is None:
        f0, pv = exchange_tracking_frames(dev, group=group)
    else:
        f0, _, _ = D.to_device_f32(frame0, ndim=(2,))
        pv, _, _ = D.to_device_f32(prev, ndim=(2,))
    T, nr = int(dev.shape[0]), len(rois)
    ra, ri = track_abs_inc(dev, f0, pv, rois, method=method, backend=backend, subpixel=subpixel, eps=eps,
                           workspace_bytes=workspace_bytes)
    f32 = lambda a: a.reshape(T, nr).astype(np.float32)  # noqa: E731
    return {"dy_abs": f32(ra[:, 0]), "dx_abs": f32(ra[:, 1]), "dy_inc": f32(ri[:, 0]), "dx_inc": f32(ri[:, 1])}


def gather_series(local: np.ndarray, *, group=None) -> np.ndarray:
    """Concatenate per-rank (T_local, ...) arrays along axis 0 on every rank (shards may differ in length).
    Two tensor collectives (lengths, then the rows padded to the longest shard): no pickling of arrays."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    arr = np.ascontiguousarray(local)
    n = torch.tensor([arr.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(v.item()) for v in sizes]
    nmax = max(max(sizes), 1)
    pad = torch.zeros((nmax,) + arr.shape[1:], dtype=torch.from_numpy(arr).dtype, device=dev)
    pad[:arr.shape[0]] = torch.from_numpy(arr).to(dev)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return np.concatenate([p[:k].cpu().numpy() for p, k in zip(parts, sizes)], axis=0)
