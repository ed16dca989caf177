"""GPU tier: deconvolve_psf (Wiener) through the C ABI vs the (unpinned) oracle restatement.
Tolerance: float32 transforms on both sides, compared normwise at 2e-5 of the frame maximum."""
import numpy as np
import pytest

from barc4dip_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pp():
    import torch

    assert torch.cuda.is_available()
    from barc4dip_amd import preprocessing

    return preprocessing


@pytest.mark.parametrize("shape,sigma", [((60, 52), 1.5), ((64, 64), (1.0, 2.0)), ((100, 37), 0.7), ((512, 512), 1.5),
                                         ((70, 258), 0.7), ((258, 258), 0.7),   # 262 = 2 * 131: DFT-matrix fallback
                                         # padded 264 = 8*3*11 / 520 = 8*5*13: the three-kernel mixed-radix route
                                         # (b4d_wiener_mr.hip) on square and both non-square orientations
                                         ((256, 256), 1.5), ((256, 512), 1.5), ((512, 256), 1.5)])
@pytest.mark.parametrize("clip", [True, False])
def test_wiener_vs_oracle(pp, shape, sigma, clip):
    from oracle import wiener_np as W

    img = synth.speckle_frame(512, 9)[:shape[0], :shape[1]].copy()
    got = pp.deconvolve_psf(img, sigma=sigma, clip=clip)
    ref = W.deconvolve_psf(img, sigma=sigma, clip=clip)
    assert got.shape == img.shape and got.dtype == np.float32
    assert float(np.max(np.abs(got - ref))) < 2e-5 * float(np.max(np.abs(img)))


def test_wiener_detector_format(pp):
    """A non-square camera format on the three-kernel route: 1072 x 1912, sigma 1.5 -> padded 1080 x 1920 (12 * 10 * 9 and
    16 * 12 * 10), and a stack call against the per-frame calls."""
    import torch
    from oracle import wiener_np as W

    img = synth.speckle_frame(2048, 21)[:1072, :1912].copy()
    ref = W.deconvolve_psf(img, sigma=1.5)
    got = pp.deconvolve_psf(img, sigma=1.5)
    assert got.shape == img.shape and got.dtype == np.float32
    assert float(np.max(np.abs(got - ref))) < 2e-5 * float(np.max(np.abs(img)))
    st = np.stack([img, img[::-1].copy(), img[:, ::-1].copy(), img * 0.5, img + 7])
    out = pp.deconvolve_psf(torch.from_numpy(st).cuda(), sigma=1.5, return_tensors=True).cpu().numpy()
    for i in range(len(st)):
        assert np.array_equal(out[i], pp.deconvolve_psf(st[i], sigma=1.5))


@pytest.mark.parametrize("clip", [True, False])
def test_cfg5_size(pp, clip):
    """BASELINE.json config 5 at its stated size: one 4096 x 4096 frame, sigma 1.5 (9 x 9 PSF, padded 4104 = 8 * 27 * 19),
    preprocessing/filters.py:233-289.  2e-5 of the data range against the float32 oracle (parity unpinned: scikit-image).
    The clip matters here: the Wiener filter overshoots the normalised range at the brightest grains."""
    import torch
    from oracle import wiener_np as W

    img = synth.speckle_frame(4096, 77)
    ref = W.deconvolve_psf(img, sigma=1.5, clip=clip)
    got = pp.deconvolve_psf(img, sigma=1.5, clip=clip)
    assert got.shape == img.shape and got.dtype == np.float32
    err = float(np.max(np.abs(got - ref))) / float(np.max(np.abs(img)))
    print(f"cfg5 4096^2 clip={clip}: max err / range = {err:.2e}")
    assert err < 2e-5
    if clip:   # says whether the clip is active on this frame (then max|restored| is exactly the frame maximum)
        print("clip active:", float(np.max(np.abs(ref))) == float(np.max(np.abs(img))))
    # a multi-frame call (several frames per launch) reproduces the single-frame result bit for bit
    dev = torch.from_numpy(np.stack([img, img[::-1].copy(), img[:, ::-1].copy()])).cuda()
    out = pp.deconvolve_psf(dev, sigma=1.5, clip=clip, return_tensors=True)
    assert np.array_equal(out[0].cpu().numpy(), got)
    assert np.array_equal(out[1].cpu().numpy(), pp.deconvolve_psf(img[::-1].copy(), sigma=1.5, clip=clip))


def test_mixed_radix_route_special_values(pp):
    """np.nanmax / np.clip semantics on the mixed-radix route (padded 264): NaN pixel -> NaN frame, all-zero and
    infinite-maximum frames -> zeros (filters.py:255-257), an odd number of frames per call."""
    from oracle import wiener_np as W

    stack = synth.speckle_stack(5, 256, seed0=90)
    stack[1, 10, 20] = np.nan
    stack[2] = 0.0
    stack[3, 100, 7] = np.inf
    got = pp.deconvolve_psf(stack, sigma=1.5)
    assert np.isnan(got[1]).all() and np.all(got[2] == 0) and np.all(got[3] == 0)
    for t in (0, 4):
        ref = W.deconvolve_psf(stack[t], sigma=1.5)
        assert float(np.max(np.abs(got[t] - ref))) < 2e-5 * float(stack[t].max())


def test_stack_balance_and_errors(pp):
    from oracle import wiener_np as W

    stack = synth.speckle_stack(3, 128, seed0=40)[:, :100, :120].copy()
    got = pp.deconvolve_psf(stack, sigma=1.5, balance=0.1)
    ref = W.deconvolve_psf(stack, sigma=1.5, balance=0.1)
    assert float(np.max(np.abs(got - ref))) < 2e-5 * float(stack.max())
    assert np.all(pp.deconvolve_psf(np.zeros((32, 32), np.float32), sigma=1.0) == 0)
    with pytest.raises(TypeError):
        pp.deconvolve_psf([[1.0]], sigma=1.0)
    with pytest.raises(ValueError):
        pp.deconvolve_psf(stack[0, 0], sigma=1.0)
    with pytest.raises(ValueError):
        pp.deconvolve_psf(stack, sigma=-1.0)
    with pytest.raises(ValueError):
        pp.deconvolve_psf(stack, sigma=1.0, method="bogus")
    with pytest.raises(ValueError):
        pp.deconvolve_psf(stack, sigma=1.0, pad_mode="edge")
    with pytest.raises(NotImplementedError):
        pp.deconvolve_psf(stack, sigma=1.0, method="uw")


def test_stack_frames_on_two_streams_keep_the_callers_order(pp):
    """A multi-frame call runs alternate frames on two plan-owned streams (include/b4d.h): each frame must equal the
    single-frame call bit for bit, on the default stream and on a side stream with consumers queued right behind it."""
    import torch

    dev = torch.device("cuda:0")
    stack = torch.from_numpy(synth.speckle_stack(5, 512, seed0=70)[:, :500, :404].copy()).to(dev)
    singles = torch.stack([pp.deconvolve_psf(stack[t], sigma=1.5, return_tensors=True) for t in range(5)])
    torch.cuda.synchronize()
    for _ in range(3):
        got = pp.deconvolve_psf(stack, sigma=1.5, return_tensors=True)
        total = got.sum(dtype=torch.float64)         # queued on the same stream right after the call
        assert torch.equal(got, singles)
        assert float(total) == float(singles.sum(dtype=torch.float64))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        scaled = stack * 2.0                          # producer queued on the side stream just before the call
        got = pp.deconvolve_psf(scaled, sigma=1.5, return_tensors=True)
        halves = got * 0.5                            # consumer queued behind it
    side.synchronize()
    ref = torch.stack([pp.deconvolve_psf(scaled[t], sigma=1.5, return_tensors=True) for t in range(5)])
    torch.cuda.synchronize()
    assert torch.equal(got, ref) and torch.equal(halves, ref * 0.5)


@pytest.mark.parametrize("shape,sigma,iters", [((60, 52), 1.5, 12), ((130, 200), (1.0, 2.0), 30), ((512, 512), 1.5, 8)])
def test_richardson_lucy_vs_oracle(pp, shape, sigma, iters):
    """method="rl" (filters.py:270-277) against the float32 oracle of the published algorithm (parity with scikit-image
    unpinned).  Both run float32 direct convolutions; summation order differs, the multiplicative iteration amplifies
    rounding mildly: 2e-5 of the data range after a few tens of iterations."""
    from barc4dip_amd import synth
    from oracle import wiener_np as W

    img = synth.speckle_frame(max(shape), 11)[:shape[0], :shape[1]]
    want = W.deconvolve_psf(img, sigma=sigma, method="rl", num_iter=iters)
    got = pp.deconvolve_psf(img, sigma=sigma, method="rl", num_iter=iters)
    assert got.dtype == np.float32 and got.shape == img.shape
    assert np.max(np.abs(got - want)) <= 2e-5 * np.max(np.abs(img))
    got_fe = pp.deconvolve_psf(img, sigma=sigma, method="rl", num_iter=3, filter_epsilon=0.05, clip=False)
    want_fe = W.deconvolve_psf(img, sigma=sigma, method="rl", num_iter=3, filter_epsilon=0.05, clip=False)
    assert np.max(np.abs(got_fe - want_fe)) <= 2e-5 * np.max(np.abs(img))
    st = np.stack([img, img[::-1].copy()])
    out2 = pp.deconvolve_psf(st, sigma=sigma, method="rl", num_iter=4)
    assert np.array_equal(out2[0], pp.deconvolve_psf(img, sigma=sigma, method="rl", num_iter=4))
    with pytest.raises(ValueError):
        pp.deconvolve_psf(img, sigma=sigma, method="rl", num_iter=0)
    with pytest.raises(NotImplementedError):
        pp.deconvolve_psf(img, sigma=sigma, method="uw")
