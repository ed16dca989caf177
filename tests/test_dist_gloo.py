"""CPU tier: the N > 1 path of the temporal statistics (frame-wise sharding + ONE all-reduce of the
float64 sums) with world_size 2 and 3 over gloo.  The device kernels are replaced by their oracle
restatement here (this tier has no GPU); the collective and the sharding are the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from barc4dip_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from barc4dip_amd.metrics.temporal import reduce_sums_cpu, shard_bounds
    from oracle import temporal_np as Tn

    stack = synth.speckle_stack(total, 64, seed0=900)
    t0, t1 = shard_bounds(total, world, rank)
    sx, sxx, n = Tn.temporal_sums(stack[t0:t1]) if t1 > t0 else (np.zeros((64, 64)), np.zeros((64, 64)), 0)
    gx, gxx, cnt = reduce_sums_cpu(sx, sxx, n)
    mean, var, con = Tn.finalize_sums(gx, gxx, cnt)
    # the reduce-scatter + all-gather route (row slices, one per rank; 64 rows do not divide by 3: the last slice is padded)
    from barc4dip_amd.metrics.temporal import scatter_reduce_sums_cpu

    smean, svar, scon, scnt = scatter_reduce_sums_cpu(sx, sxx, n, Tn.finalize_sums)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), mean=mean, var=var, con=con, cnt=cnt, smean=smean, svar=svar, scon=scon, scnt=scnt)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 11), (3, 7)])
def test_sharded_temporal_stats_equal_single_rank(tmp_path, world, total):
    from oracle import temporal_np as Tn

    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    stack = synth.speckle_stack(total, 64, seed0=900)
    rm, rv, rc = Tn.temporal_stats(stack)
    for r in range(world):
        g = np.load(tmp_path / f"r{r}.npz")
        assert float(g["cnt"]) == total
        np.testing.assert_allclose(g["mean"], rm, rtol=1e-14)
        np.testing.assert_allclose(g["var"], rv, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(g["con"], rc, rtol=1e-9)
        # same sums, same finalisation, float32 on the wire: the scatter route equals the all-reduce route rounded to float32
        assert float(g["scnt"]) == total
        for k in ("mean", "var", "con"):
            assert np.array_equal(g["s" + k], g[k].astype(np.float32)), k
    a, b = np.load(tmp_path / "r0.npz"), np.load(tmp_path / f"r{world - 1}.npz")
    assert np.array_equal(a["mean"], b["mean"]) and np.array_equal(a["var"], b["var"])   # every rank holds the same bits


def _halo_worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from barc4dip_amd.metrics.sharded import exchange_tracking_frames, gather_series, shard_bounds

    stack = synth.speckle_stack(total, 32, seed0=40)
    t0, t1 = shard_bounds(total, world, rank)
    f0, prev = exchange_tracking_frames(torch.from_numpy(stack[t0:t1]))
    series = gather_series(np.arange(t0, t1, dtype=np.float32)[:, None] * np.ones((1, 3), np.float32))
    np.savez(os.path.join(out_dir, f"h{rank}.npz"), f0=f0.numpy(), prev=prev.numpy(), t0=t0, series=series)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 7), (3, 10)])
def test_tracking_halo_and_frame0_broadcast(tmp_path, world, total):
    """every rank ends up with global frame 0 and the frame just before its shard; the gathered series
    is the global frame order on every rank."""
    mp.spawn(_halo_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    stack = synth.speckle_stack(total, 32, seed0=40)
    for r in range(world):
        g = np.load(tmp_path / f"h{r}.npz")
        t0 = int(g["t0"])
        assert np.array_equal(g["f0"], stack[0])
        assert np.array_equal(g["prev"], stack[max(t0 - 1, 0)])
        assert np.array_equal(g["series"][:, 0], np.arange(total, dtype=np.float32))
