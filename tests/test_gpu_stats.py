"""GPU tier: reduction kernels (moments, Sobel/Laplace, temporal sums) vs the oracle."""
import numpy as np
import pytest

from barc4dip_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    import torch

    assert torch.cuda.is_available()
    from barc4dip_amd.metrics import kernels

    return kernels


def test_moments_vs_oracle(K):
    from oracle import metrics_np as M

    frames = np.stack([synth.speckle_frame(512, 5)[:500, :404], synth.speckle_frame(512, 6)[:500, :404] * 60.0])
    frames[1, 3, 7] = np.nan
    frames[1, 100, 9] = np.inf
    frames[0, :4, :4] = 0.0
    out = K.moments_batch(frames, eps=1e-6, saturation=3000.0).cpu().numpy()
    for b in range(2):
        ref = M.distribution_moments(frames[b], saturation_value=3000.0)
        n, mean, m2, m3, m4, nz, ns = out[b, :7]
        assert n == np.isfinite(frames[b]).sum()
        assert mean == pytest.approx(ref["mean"], rel=1e-12)
        var = m2 / n
        assert np.sqrt(var) == pytest.approx(ref["std"], rel=1e-12)
        assert (m3 / n) / var ** 1.5 == pytest.approx(ref["skewness"], rel=1e-10)
        assert (m4 / n) / var ** 2 - 3.0 == pytest.approx(ref["kurtosis"], rel=1e-10)
        assert nz / n == ref["frac_zero"] and ns / n == ref["frac_sat"]


@pytest.mark.parametrize("shape", [(171, 170), (512, 512), (64, 700)])
def test_sobel_laplace_vs_oracle(K, shape):
    from oracle import metrics_np as M

    frames = np.stack([synth.speckle_frame(1024, 9)[:shape[0], :shape[1]], synth.speckle_frame(1024, 10)[:shape[0], :shape[1]]])
    out = K.sobel_laplace_batch(frames).cpu().numpy()
    for b in range(2):
        t = M.tenengrad(frames[b])
        assert out[b, 0] == pytest.approx(t["ex"], rel=1e-12)
        assert out[b, 1] == pytest.approx(t["ey"], rel=1e-12)
        lv = M.laplacian_variance(frames[b])
        assert out[b, 3] - out[b, 2] ** 2 == pytest.approx(lv, rel=1e-10)


def test_temporal_stats_vs_oracle():
    from barc4dip_amd.metrics import temporal_stats
    from oracle import temporal_np as Tn

    stack = synth.speckle_stack(64, 256, seed0=300)          # the cfg4 parity case: (64, 256, 256)
    mean, var, con = temporal_stats(stack, chunk=24)         # ragged chunks on purpose
    rm, rv, rc = Tn.temporal_stats(stack)
    np.testing.assert_allclose(mean, rm, rtol=2e-7)
    np.testing.assert_allclose(var, rv, rtol=2e-6)
    np.testing.assert_allclose(con, rc, rtol=2e-6)
    # exactness of the float64 sums: integers (Poisson counts) sum exactly
    import torch
    from barc4dip_amd.metrics import kernels as K

    dev = torch.from_numpy(stack).cuda()
    s = torch.zeros((2, 256, 256), dtype=torch.float64, device="cuda")
    K.temporal_accumulate(dev, s[0], s[1])
    sx, sxx, n = Tn.temporal_sums(stack)
    assert np.array_equal(s[0].cpu().numpy(), sx) and np.array_equal(s[1].cpu().numpy(), sxx)


def test_streamed_ingest_equals_resident(tmp_path):
    """iter_device_chunks / temporal_stats_streamed (pinned double buffering, side copy stream) over a memory-mapped
    uint16 .npy file give the bits of the device-resident path; ragged last chunk, chunk > T and chunk = 1 included."""
    import torch

    from barc4dip_amd import ingest, synth
    from barc4dip_amd.metrics import temporal_stats

    stack = np.stack([synth.speckle_frame(128, 400 + i) for i in range(11)]).astype(np.uint16)
    path = tmp_path / "stack.npy"
    np.save(path, stack)
    mm = np.load(path, mmap_mode="r")
    want = temporal_stats(stack.astype(np.float32))
    for chunk in (4, 1, 64):
        got = ingest.temporal_stats_streamed(mm, chunk_frames=chunk)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        seen = [c.clone() for c in ingest.iter_device_chunks(mm, chunk)]
        assert torch.equal(torch.cat(seen), torch.from_numpy(stack.astype(np.float32)).cuda())
    with pytest.raises(ValueError):
        list(ingest.iter_device_chunks(stack[0], 4))


@pytest.mark.parametrize("shape", [(5, 33, 31), (7, 101, 127), (3, 64, 66), (9, 1, 5)])
def test_temporal_stats_any_frame_shape(shape):
    """The reference's data.mean(axis=0) (io/rw.py:129-132) takes any frame shape: odd pixel counts and frames that
    do not start on 16-byte boundaries take the dword kernel (b4d_temporal_accumulate_range)."""
    import torch

    from barc4dip_amd.metrics import temporal_stats
    from oracle import temporal_np as Tn

    rng = np.random.default_rng(shape[1] * 100 + shape[2])
    stack = rng.poisson(700.0, size=shape).astype(np.float32)
    for chunk in (2, 1024):
        got = temporal_stats(stack, chunk=chunk)
        for g, r in zip(got, Tn.temporal_stats(stack)):
            assert g.shape == shape[1:] and g.dtype == np.float32
            np.testing.assert_allclose(g, r, rtol=2e-6, atol=1e-6)
    view = torch.from_numpy(np.concatenate([np.zeros(1, np.float32), stack.ravel()])).cuda()[1:].view(shape)   # 4-byte offset
    got2 = temporal_stats(view, return_tensors=True)
    assert all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(got2, got))


def test_temporal_stats_packed_allreduce_and_row_block_overlap():
    """cfg4's collective on RCCL (a 1-rank group on this one-GPU box: same calls, same streams): the count rides in the
    float64 buffer of the sums (ONE all-reduce), and with overlap_chunks > 1 every row block is reduced on a side stream
    while the next one accumulates; both must reproduce the plain single-process result bit for bit."""
    import os

    import torch
    import torch.distributed as dist

    from barc4dip_amd import synth
    from barc4dip_amd.metrics import temporal_stats

    stack = torch.from_numpy(synth.speckle_stack(12, 256, seed0=3)[:, :250, :252].copy()).cuda()
    plain = [x.cpu().numpy() for x in temporal_stats(stack, return_tensors=True)]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29871")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        tm = {}
        one = [x.cpu().numpy() for x in temporal_stats(stack, return_tensors=True, timings=tm)]
        assert "allreduce_ms" in tm and tm["allreduce_ms"] >= 0.0
        blocks = [x.cpu().numpy() for x in temporal_stats(stack, return_tensors=True, overlap_chunks=5, chunk=5)]
        # SURVEY.md section 8e's alternative: reduce-scatter of row slices, local finalisation, all-gather of the float32 maps
        ts = {}
        scat = [x.cpu().numpy() for x in temporal_stats(stack, return_tensors=True, collective="reduce_scatter", timings=ts)]
        assert ts["reduce_scatter_ms"] >= 0.0 and ts["all_gather_ms"] >= 0.0
        with pytest.raises(ValueError):
            temporal_stats(stack, collective="ring")
    finally:
        if created:
            dist.destroy_process_group()
    for a, b, c, d in zip(plain, one, blocks, scat):
        assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d, equal_nan=True)


def test_reduction_entry_points_on_two_streams(K):
    """The second-stage reduction scratch is per stream (csrc/b4d_stats.hip: get_scratch): calls queued on two torch
    streams, interleaved without any host synchronisation in between, must give the serial results."""
    import torch

    a = torch.from_numpy(synth.speckle_stack(6, 256, seed0=11)).cuda()
    b = torch.from_numpy(synth.speckle_stack(6, 256, seed0=29)[:, ::-1].copy()).cuda()
    want_a, want_b = K.moments_batch(a).cpu().numpy(), K.moments_batch(b).cpu().numpy()
    want_sa, want_sb = K.sobel_laplace_batch(a).cpu().numpy(), K.sobel_laplace_batch(b).cpu().numpy()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(5):
        with torch.cuda.stream(s1):
            ga, gsa = K.moments_batch(a), K.sobel_laplace_batch(a)
        with torch.cuda.stream(s2):
            gb, gsb = K.moments_batch(b), K.sobel_laplace_batch(b)
        s1.synchronize()
        s2.synchronize()
        assert np.array_equal(ga.cpu().numpy(), want_a) and np.array_equal(gb.cpu().numpy(), want_b)
        assert np.array_equal(gsa.cpu().numpy(), want_sa) and np.array_equal(gsb.cpu().numpy(), want_sb)
