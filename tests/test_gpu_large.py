"""GPU tier: stacks beyond 2^31 / 2^32 elements (cfg4 hands 1024 frames of 2048^2 = 2^32 floats to one GPU):
every frame offset on the path has to be 64-bit.  Device-generated inputs with closed-form answers; ~50 GiB of HBM."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch

    assert torch.cuda.is_available()
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2**30:
        pytest.skip("needs > 100 GiB of device memory")
    return torch


def test_temporal_stats_beyond_2_32_elements(torch_mod):
    """x[t, p] = (t mod 5) + (p mod 3): mean_t and var_t are known in closed form; the last frame alone crosses 2^32."""
    torch = torch_mod
    from barc4dip_amd.metrics.temporal import temporal_stats

    T, n = 1025, 2048                      # 1025 * 2^22 > 2^32 elements (16.02 GiB)
    dev = torch.device("cuda:0")
    pix = (torch.arange(n * n, device=dev, dtype=torch.int64) % 3).to(torch.float32).reshape(n, n)
    stack = torch.empty((T, n, n), dtype=torch.float32, device=dev)
    for t in range(T):
        torch.add(pix, float(t % 5), out=stack[t])
    stack[T - 1].add_(1000.0)              # a mark in the last frame: dropped or misplaced frames change every statistic
    got = temporal_stats(stack, chunk=2048, return_tensors=True)       # ONE launch over the whole > 2^32-element stack
    ref = temporal_stats(stack, chunk=64, return_tensors=True)
    tt = np.arange(T) % 5
    base = tt.astype(np.float64)
    base[-1] += 1000.0
    m0, v0 = base.mean(), base.var()
    want_mean = (pix.double() + m0).cpu().numpy()
    for g, r in zip(got, ref):             # (mean, var, contrast): chunking does not change a bit (exact float64 sums)
        assert torch.equal(g, r)
    np.testing.assert_allclose(got[0].cpu().numpy(), want_mean, rtol=1e-6)
    np.testing.assert_allclose(got[1].cpu().numpy(), np.full((n, n), v0), rtol=1e-6)
    del stack, got, ref
    torch.cuda.empty_cache()


def test_pipeline_frame_offsets_beyond_2_31_elements(torch_mod):
    """FFT -> PSD -> autocorrelation over 516 frames of 2048^2 (> 2^31 elements in and out): frames 0, 255, 512 and 515
    hold a speckle frame, the rest are zero; each marked frame must come out bit-identical to the frame processed alone."""
    torch = torch_mod
    from barc4dip_amd import synth
    from barc4dip_amd.signal.corr import psd_autocorr2d_stack

    T, n = 516, 2048
    dev = torch.device("cuda:0")
    marks = {0: 11, 255: 12, 512: 13, 515: 14}
    stack = torch.zeros((T, n, n), dtype=torch.float32, device=dev)
    for t, seed in marks.items():
        stack[t] = torch.from_numpy(synth.speckle_frame(n, seed)).to(dev)
    psd, ac = psd_autocorr2d_stack(stack, return_tensors=True)
    for t in marks:
        p1, a1 = psd_autocorr2d_stack(stack[t:t + 1].clone(), return_tensors=True)
        assert torch.equal(psd[t], p1[0]), t
        assert torch.equal(ac[t], a1[0]), t
        assert float(ac[t, n // 2, n // 2]) == 1.0
    for t in (1, 256, 511, 514):           # zero frames: PSD 0, autocorrelation stays 0 (no peak to normalise by)
        assert float(psd[t].abs().max()) == 0.0 and float(ac[t].abs().max()) == 0.0, t
    del stack, psd, ac
    torch.cuda.empty_cache()
