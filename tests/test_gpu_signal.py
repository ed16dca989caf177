"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the NumPy oracle.

Tolerances (stated per SURVEY.md §8d): FFT/PSD/autocorrelation values are float32 on the device
and are compared NORMWISE with the float64 oracle: max|got-ref| <= 1e-5 * max|ref|.  Index
outputs (arg-max of the autocorrelation) are exact, and the peak value is exactly 1.0.
"""
import os

import numpy as np
import pytest

from barc4dip_amd import synth

pytestmark = pytest.mark.gpu

TOL = 1e-5


def nerr(got, ref):
    return float(np.max(np.abs(np.asarray(got) - ref)) / np.max(np.abs(ref)))


@pytest.fixture(scope="module")
def gs():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from barc4dip_amd import _ffi, signal

    lib = _ffi.load_library()
    assert lib.b4d_missing_symbols == ()
    return signal


def frame(ny, nx, seed=1234, dtype=np.float32):
    n = max(ny, nx)
    return np.ascontiguousarray(synth.speckle_frame(n, seed)[:ny, :nx]).astype(dtype)


@pytest.mark.parametrize("shape", [(64, 64), (128, 256), (256, 64), (512, 512), (1024, 512), (512, 2048), (2048, 2048),
                                   (4096, 1024)])
def test_fft_psd_autocorr_vs_oracle(gs, shape):
    from oracle import signal_np as S

    img = frame(*shape)
    ref64 = img.astype(np.float64)
    F, fx, fy = gs.fft2d(img, dx=0.5, dy=2.0)
    Fr, fxr, fyr = S.fft2d(ref64, dx=0.5, dy=2.0)
    assert F.shape == shape and F.dtype == np.complex64
    assert nerr(F, Fr) < TOL
    np.testing.assert_array_equal(fx, fxr)
    np.testing.assert_array_equal(fy, fyr)
    # error of the reference's own float32 path, for the record: ours must not be worse than 4x that
    e32 = nerr(S.fft2d(img)[0], Fr)
    assert nerr(F, Fr) <= max(4 * e32, 2e-7)

    for kw in ({}, dict(dx=0.5, dy=2.0), dict(scale=False)):
        P = gs.psd2d(img, **kw)[0]
        Pr = S.psd2d(ref64, **kw)[0]
        assert P.dtype == np.float32
        assert nerr(P, Pr) < TOL
        cy, cx = shape[0] // 2, shape[1] // 2
        Pm, Prm = P.astype(np.float64), Pr.copy()
        Pm[cy, cx] = Prm[cy, cx] = 0.0       # without the DC bin the bound is much tighter in l2
        assert np.linalg.norm(Pm - Prm) / np.linalg.norm(Prm) < TOL

    ac, xl, yl = gs.autocorr2d(img)
    acr, xlr, ylr = S.autocorr2d(ref64)
    assert ac.dtype == np.float64 and ac.shape == shape
    assert float(np.max(np.abs(ac - acr))) < TOL           # max|ref| == 1
    assert np.unravel_index(int(np.argmax(ac)), ac.shape) == (shape[0] // 2, shape[1] // 2)
    assert ac[shape[0] // 2, shape[1] // 2] == 1.0
    np.testing.assert_array_equal(xl, xlr)
    np.testing.assert_array_equal(yl, ylr)


@pytest.mark.parametrize("rm", [True, False])
@pytest.mark.parametrize("st", [True, False])
@pytest.mark.parametrize("nm", ["peak", "none"])
def test_autocorr_options(gs, rm, st, nm):
    from oracle import signal_np as S

    img = frame(512, 1024, seed=7)
    got = gs.autocorr2d(img, remove_mean=rm, standardize=st, normalize=nm)[0]
    ref = S.autocorr2d(img.astype(np.float64), remove_mean=rm, standardize=st, normalize=nm)[0]
    assert nerr(got, ref) < TOL


def test_float64_input_and_calibrated_axes(gs):
    from oracle import signal_np as S

    img = frame(512, 512, seed=3, dtype=np.float64)
    x = np.linspace(0.0, 51.1, 512)
    y = np.linspace(10.0, 10.0 + 2 * 511, 512)
    P, fx, fy = gs.psd2d(img, x=x, y=y)
    Pr, fxr, fyr = S.psd2d(img, x=x, y=y)
    assert P.dtype == np.float64
    assert nerr(P, Pr) < TOL
    np.testing.assert_array_equal(fx, fxr)
    np.testing.assert_array_equal(fy, fyr)
    F = gs.fft2d(img)[0]
    assert F.dtype == np.complex128 and nerr(F, S.fft2d(img)[0]) < TOL


def test_errors(gs):
    with pytest.raises(ValueError):
        gs.fft2d(np.zeros(16, dtype=np.float32))
    with pytest.raises(ValueError):
        gs.psd2d(np.zeros((512, 512), dtype=np.float32), x=np.arange(512.0))
    with pytest.raises(ValueError):
        gs.psd2d(np.zeros((512, 512), dtype=np.float32), dx=-1.0)
    with pytest.raises(ValueError):
        gs.autocorr2d(np.zeros((512, 512), dtype=np.float32), normalize="bogus")
    with pytest.raises(ValueError):
        gs.autocorr2d(np.zeros((4, 512, 512), dtype=np.float32))
    with pytest.raises(NotImplementedError):           # no CPU fallback for sizes without a plan
        gs.psd2d(np.zeros((4099, 64), dtype=np.float32))               # prime side beyond the Bluestein range (4096)
    with pytest.raises(NotImplementedError):
        gs.autocorr2d(np.zeros((64, 9000), dtype=np.float32))
    with pytest.raises(NotImplementedError):
        gs.psd2d(np.zeros((512, 512), dtype=np.complex64))             # PSD / correlations take real frames only


def test_stack_equals_frames_and_chunking(gs):
    import torch

    from barc4dip_amd import _ffi
    import ctypes as C

    stack = synth.speckle_stack(5, 512, seed0=50)
    psd, ac = gs.psd_autocorr2d_stack(stack)
    for t in range(5):
        np.testing.assert_array_equal(psd[t], gs.psd2d(stack[t])[0])
        np.testing.assert_array_equal(ac[t], gs.autocorr2d(stack[t])[0].astype(np.float32))
    # chunk size must not change a single bit (ragged last chunk included)
    dev = torch.from_numpy(stack).cuda()
    outs = []
    for chunk in (1, 2, 5, 8):
        pl = _ffi.Plan(512, 512, chunk)
        p = torch.empty_like(dev)
        a = torch.empty_like(dev)
        _ffi.check(_ffi.lib().b4d_psd_autocorr2d(pl.handle, C.c_void_p(dev.data_ptr()), 5, C.c_void_p(p.data_ptr()),
                                                 1.0, C.c_void_p(a.data_ptr()), 3, _ffi.stream_ptr()))
        torch.cuda.synchronize()
        outs.append((p.cpu().numpy(), a.cpu().numpy()))
        pl.close()
    for p, a in outs[1:]:
        np.testing.assert_array_equal(p, outs[0][0])
        np.testing.assert_array_equal(a, outs[0][1])


def test_plan_tune_keeps_results_and_a_working_plan(gs):
    """b4d_plan_tune exchanges the plan's workspace for the fastest of several allocations: outputs after the call, and of
    later calls on the tuned plan, are bit-identical to an untuned plan's; general-length plans take it as a no-op."""
    import torch

    from barc4dip_amd import _ffi
    import ctypes as C

    stack = synth.speckle_stack(6, 512, seed0=77)
    dev = torch.from_numpy(stack).cuda()

    def call(pl, p, a):
        _ffi.check(_ffi.lib().b4d_psd_autocorr2d(pl.handle, C.c_void_p(dev.data_ptr()), 6, C.c_void_p(p.data_ptr()), 0.5,
                                                 C.c_void_p(a.data_ptr()), 3, _ffi.stream_ptr()))
        torch.cuda.synchronize()

    ref_p, ref_a = torch.empty_like(dev), torch.empty_like(dev)
    pl0 = _ffi.Plan(512, 512, 4)
    call(pl0, ref_p, ref_a)
    pl0.close()
    pl = _ffi.Plan(512, 512, 4)
    p, a = torch.full_like(dev, -1.0), torch.full_like(dev, -1.0)
    best, worst = pl.tune(dev, p, a, psd_scale=0.5, flags=3, candidates=4)
    assert 0.0 < best <= worst
    assert torch.equal(p, ref_p) and torch.equal(a, ref_a)
    p.fill_(-1.0)
    a.fill_(-1.0)
    call(pl, p, a)
    assert torch.equal(p, ref_p) and torch.equal(a, ref_a)
    with pytest.raises(_ffi.B4DError):
        pl.tune(dev, p, a, candidates=1)
    pl.close()
    g = _ffi.Plan(300, 300, 2)   # DFT-matrix plan: nothing to tune, nothing reported
    d3 = torch.zeros((2, 300, 300), dtype=torch.float32, device="cuda")
    assert g.tune(d3, torch.empty_like(d3), None) == (0.0, 0.0)
    g.close()


def test_full_size_properties(gs):
    """Size-independent properties at the benchmark size (2048^2, 8-frame stack)."""
    import torch

    T, n = 8, 2048
    dev = synth.speckle_stack_device(T, n, seed0=99)
    psd, ac = gs.psd_autocorr2d_stack(dev, scale=False, return_tensors=True)
    x = dev.double()
    # Parseval: sum |F|^2 = N^2 * sum x^2
    lhs = psd.double().sum(dim=(1, 2))
    rhs = (x * x).sum(dim=(1, 2)) * (n * n)
    assert float(((lhs - rhs).abs() / rhs).max()) < 1e-6
    # DC bin = (sum x)^2
    dc = psd[:, n // 2, n // 2].double()
    assert float(((dc - x.sum(dim=(1, 2)) ** 2).abs() / dc).max()) < 1e-6
    # Hermitian symmetry of the PSD and evenness of the autocorrelation: out[-k] == out[k]
    def mirror(a):
        return torch.roll(torch.flip(a, dims=(1, 2)), shifts=(1, 1), dims=(1, 2))
    mp = mirror(psd)
    assert float(((mp - psd)[:, 1:, 1:].abs().amax(dim=(1, 2)) / psd.amax(dim=(1, 2))).max()) < 1e-6
    keep = torch.ones(n, dtype=torch.bool, device=psd.device)
    keep[0] = keep[n // 2] = False           # kx = nx/2 and kx = 0 columns come from their own transforms
    assert torch.equal(mp[:, 1:][:, :, keep], psd[:, 1:][:, :, keep])   # elsewhere the mirror is bitwise
    assert float((mirror(ac) - ac).abs().max()) < 2e-6
    # peak: exactly one, at the centre, strictly the maximum
    assert torch.all(ac[:, n // 2, n // 2] == 1.0)
    flat = ac.reshape(T, -1)
    assert torch.all(flat.argmax(dim=1) == (n // 2) * n + n // 2)
    # mean removal: the autocorrelation sums to ~0 (DC bin zeroed)
    assert float(ac.double().sum(dim=(1, 2)).abs().max()) < 1e-2 * n
    # circular shift invariance of PSD and autocorrelation
    sh = torch.roll(dev, shifts=(37, -501), dims=(1, 2))
    psd2, ac2 = gs.psd_autocorr2d_stack(sh, scale=False, return_tensors=True)
    assert float(((psd2 - psd).abs().amax(dim=(1, 2)) / psd.amax(dim=(1, 2))).max()) < 1e-6
    assert float((ac2 - ac).abs().max()) < 1e-5
    # linearity of fft2d
    F1 = gs.fft2d_stack(dev[:1], return_tensors=True)
    F2 = gs.fft2d_stack(dev[1:2], return_tensors=True)
    F12 = gs.fft2d_stack(dev[:1] * 2 - dev[1:2], return_tensors=True)
    assert float(((F12 - (2 * F1 - F2)).abs().max() / F12.abs().max())) < 1e-6
    # ifft2d(fft2d(x)) ~ x (host inverse)
    xr = gs.fft.ifft2d(F1[0].cpu().numpy())
    assert float(np.max(np.abs(xr.real - dev[0].cpu().numpy())) / float(dev[0].max())) < 1e-5


def test_golden_reference_vectors_64(gs, golden):
    """Committed outputs of the REAL reference (tests/golden/signal_small.npz, f32_64 case)."""
    g = golden("signal_small.npz")
    a, b = g["f32_64/a"], g["f32_64/b"]
    assert nerr(gs.fft2d(a, dx=0.5, dy=2.0)[0], g["f32_64/fft2d"]) < TOL
    assert nerr(gs.psd2d(a)[0], g["f32_64/psd2d"]) < TOL
    assert nerr(gs.psd2d(a, dx=0.5, dy=2.0)[0], g["f32_64/psd2d_cal"]) < TOL
    assert nerr(gs.psd2d(a, scale=False)[0], g["f32_64/psd2d_noscale"]) < TOL
    assert nerr(gs.autocorr2d(a)[0], g["f32_64/autocorr2d_rm1_st0_peak"]) < TOL
    assert nerr(gs.autocorr2d(a, standardize=True, normalize="none")[0], g["f32_64/autocorr2d_rm1_st1_none"]) < TOL
    assert nerr(gs.xcorr2d(a, b)[0], np.real(g["f32_64/xcorr2d_rm1_st0_peak"])) < TOL
    assert nerr(gs.xcorr2d(a, b, standardize=True, normalize="none")[0], np.real(g["f32_64/xcorr2d_rm1_st1_none"])) < TOL


@pytest.mark.parametrize("shape", [(171, 170), (228, 227), (100, 100), (33, 45), (32, 512), (2, 3), (228, 228), (170, 171),
                                   (171, 171), (228, 170)])
def test_general_lengths_vs_oracle(gs, shape):
    """Any ny, nx <= 512 (the aggregators' 170/171- and 227/228-pixel tiles): DFT-matrix plans, or -- when both sides are compiled
    mixed-radix lengths (228 = 4*3*19, 171 = 3*3*19 (odd), 170 = 5*2*17) -- the three-pass mixed-radix route."""
    from oracle import signal_np as S

    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    img = (rng.poisson(200.0, size=shape) + 5 * rng.random(shape)).astype(np.float32)
    b = (np.roll(img, (1, -1), axis=(0, 1)) * 0.9 + rng.random(shape)).astype(np.float32)
    r64 = img.astype(np.float64)
    assert nerr(gs.fft2d(img)[0], S.fft2d(r64)[0]) < TOL
    assert nerr(gs.psd2d(img, dx=0.5, dy=2.0)[0], S.psd2d(r64, dx=0.5, dy=2.0)[0]) < TOL
    for kw in (dict(), dict(remove_mean=False, normalize="none"), dict(standardize=True, normalize="none")):
        assert nerr(gs.autocorr2d(img, **kw)[0], S.autocorr2d(r64, **kw)[0]) < TOL
        assert nerr(gs.xcorr2d(img, b, **kw)[0], np.real(S.xcorr2d(r64, b.astype(np.float64), **kw)[0])) < TOL
    ac = gs.autocorr2d(img)[0]
    assert ac[shape[0] // 2, shape[1] // 2] == 1.0 and int(np.argmax(ac)) == (shape[0] // 2) * shape[1] + shape[1] // 2


@pytest.mark.parametrize("shape", [(600, 600), (720, 1280), (1000, 2048), (513, 300), (2160, 2560), (264, 520), (520, 264)])
def test_large_general_lengths_vs_oracle(gs, shape):
    """Sides beyond the DFT-matrix range that split as 2^k * A * B (detector formats such as 2560 x 2160): fused
    in-LDS mixed-radix transform, rows / transpose / columns.  Same 1e-5 bar as the power-of-two kernels."""
    from oracle import signal_np as S

    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    img = (rng.poisson(200.0, size=shape) + 5 * rng.random(shape)).astype(np.float32)
    b = (np.roll(img, (2, -3), axis=(0, 1)) * 0.9 + rng.random(shape)).astype(np.float32)
    r64 = img.astype(np.float64)
    assert nerr(gs.fft2d(img)[0], S.fft2d(r64)[0]) < TOL
    assert nerr(gs.psd2d(img, dx=0.5, dy=2.0)[0], S.psd2d(r64, dx=0.5, dy=2.0)[0]) < TOL
    assert nerr(gs.autocorr2d(img)[0], S.autocorr2d(r64)[0]) < TOL
    assert nerr(gs.xcorr2d(img, b)[0], np.real(S.xcorr2d(r64, b.astype(np.float64))[0])) < TOL
    ac = gs.autocorr2d(img)[0]
    assert ac[shape[0] // 2, shape[1] // 2] == 1.0 and int(np.argmax(ac)) == (shape[0] // 2) * shape[1] + shape[1] // 2
    st = np.stack([img, b, img[::-1].copy()])
    p3 = gs.psd2d_stack(st)
    f3 = gs.fft2d_stack(st)
    for i in range(3):
        assert np.array_equal(p3[i], gs.psd2d(st[i])[0])
        assert np.array_equal(f3[i], gs.fft2d(st[i])[0])
    # full spectrum of a real frame: Hermitian in the shifted layout, F[-ky, -kx] = conj(F[ky, kx]) bit for bit (one value, two stores)
    F = gs.fft2d(img)[0]
    cy, cx = shape[0] // 2, shape[1] // 2
    ys = (2 * cy - np.arange(shape[0])) % shape[0] if shape[0] % 2 == 0 else shape[0] - 1 - np.arange(shape[0])
    xs = (2 * cx - np.arange(shape[1])) % shape[1] if shape[1] % 2 == 0 else shape[1] - 1 - np.arange(shape[1])
    inner = np.ix_(np.arange(shape[0])[1:], np.arange(shape[1])[1:])   # row / column 0 of an even side hold the unpaired Nyquist bins
    assert nerr(F[inner], np.conj(F[np.ix_(ys, xs)])[inner]) < 1e-6


@pytest.mark.parametrize("shape", [(480, 640), (540, 960), (576, 768), (512, 800), (1000, 1024), (1080, 1920), (1200, 1600),
                                   (1216, 1936), (1440, 2304), (1536, 2048), (1944, 2592), (2400, 3200), (2448, 3072),
                                   (3000, 4096), (3648, 3840)])
def test_detector_formats_mixed_radix(gs, shape):
    """Every length of the mixed-radix table (csrc/b4d_wiener_mr.hip: B4D_WMR_LENGTHS) that the tests above do not reach, once
    as a row length or a column length: camera / detector sides (1080 x 1920, 1200 x 1600, 1216 x 1936, 1944 x 2592, 2448 ...)
    and the powers of two next to them.  fft2d, psd2d, autocorr2d, xcorr2d against the float64 oracle at the 1e-5 bar."""
    from oracle import signal_np as S

    rng = np.random.default_rng(shape[0] * 5 + shape[1])
    img = (rng.poisson(150.0, size=shape) + 3 * rng.random(shape)).astype(np.float32)
    b = (np.roll(img, (-4, 7), axis=(0, 1)) * 1.1 + rng.random(shape)).astype(np.float32)
    r64 = img.astype(np.float64)
    assert nerr(gs.fft2d(img)[0], S.fft2d(r64)[0]) < TOL
    assert nerr(gs.psd2d(img)[0], S.psd2d(r64)[0]) < TOL
    ac = gs.autocorr2d(img)[0]
    assert nerr(ac, S.autocorr2d(r64)[0]) < TOL
    assert ac[shape[0] // 2, shape[1] // 2] == 1.0 and int(np.argmax(ac)) == (shape[0] // 2) * shape[1] + shape[1] // 2
    xc = gs.xcorr2d(img, b)[0]
    assert nerr(xc, np.real(S.xcorr2d(r64, b.astype(np.float64))[0])) < TOL
    assert np.unravel_index(int(np.argmax(np.abs(xc))), shape) == (shape[0] // 2 + 4, shape[1] // 2 - 7)


@pytest.mark.parametrize("name", ["f64_24x32", "f32_32x16", "f64_17x23"])
def test_golden_reference_vectors_small(gs, golden, name):
    """The small dense cases captured from the REAL reference (odd, non-square, float64 inputs)."""
    g = golden("signal_small.npz")
    a, b = g[f"{name}/a"], g[f"{name}/b"]
    assert nerr(gs.fft2d(a, dx=0.5, dy=2.0)[0], g[f"{name}/fft2d"]) < TOL
    assert nerr(gs.psd2d(a)[0], g[f"{name}/psd2d"]) < TOL
    assert nerr(gs.psd2d(a, scale=False)[0], g[f"{name}/psd2d_noscale"]) < TOL
    for rm in (True, False):
        for st in (True, False):
            for nm in ("peak", "none"):
                tag = f"rm{int(rm)}_st{int(st)}_{nm}"
                kw = dict(remove_mean=rm, standardize=st, normalize=nm)
                assert nerr(gs.autocorr2d(a, **kw)[0], g[f"{name}/autocorr2d_{tag}"]) < TOL, tag
                assert nerr(gs.xcorr2d(a, b, **kw)[0], np.real(g[f"{name}/xcorr2d_{tag}"])) < TOL, tag


@pytest.mark.parametrize("name", ["f64_24x32", "f32_32x16", "f64_17x23", "f32_64"])
def test_xcorr2d_return_type_follows_the_reference(gs, golden, name):
    """signal/corr.py:41-42, 242: np.real_if_close(tol=1000) leaves the reference's xcorr2d complex128 whenever the
    rounding noise of its imaginary part exceeds 2.2e-13 absolute, i.e. for |corr| beyond a few thousand.  The device
    result has an exactly-zero imaginary part and takes the reference's dtype from the modelled noise level
    (barc4dip_amd/signal/corr.py); cases within 2.5 x of the modelled bound (max|corr| = 4000) are the reference's own
    rounding lottery and are not asserted.  np.argmax (lexicographic on complex) finds the same element either way."""
    g = golden("signal_small.npz")
    a, b = g[f"{name}/a"], g[f"{name}/b"]
    checked = 0
    for rm in (True, False):
        for st in (True, False):
            for nm in ("peak", "none"):
                tag, raw_tag = f"rm{int(rm)}_st{int(st)}_{nm}", f"rm{int(rm)}_st{int(st)}_none"
                if f"{name}/xcorr2d_{tag}" not in g.files:
                    continue
                want = g[f"{name}/xcorr2d_{tag}"]
                got = gs.xcorr2d(a, b, remove_mean=rm, standardize=st, normalize=nm)[0]
                assert got.dtype in (np.float64, np.complex128)
                if np.iscomplexobj(got):
                    assert np.all(got.imag == 0)
                assert int(np.argmax(got)) == int(np.argmax(want)), tag
                if f"{name}/xcorr2d_{raw_tag}" in g.files:
                    raw = float(np.max(np.abs(g[f"{name}/xcorr2d_{raw_tag}"])))
                    if raw > 10000 or raw < 1600:
                        assert got.dtype == want.dtype, (tag, raw)
                        checked += 1
    assert checked > 0 or name == "f32_64"


def test_xcorr2d_return_type_band_is_pinned_by_reference_data(gs, golden):
    """tests/golden/xcorr_dtype.npz: the real reference on 19 shapes x 48 amplitudes returned complex128 from max|corr| = 1615 on
    (smallest) and float64 up to 10393 (largest); outside that band its dtype depends on neither shape nor data, and there the
    device path must return the same type (its rule: complex128 beyond 4000, inside the band).  np.argmax agrees everywhere."""
    from oracle import signal_np as S

    g = golden("xcorr_dtype.npz")
    lo, hi = (float(v) for v in g["band"])
    checked = 0
    for k, (shape, seed) in enumerate(zip(g["shapes"], g["seeds"])):
        rng = np.random.default_rng(int(seed))
        a = rng.poisson(50.0, size=tuple(int(v) for v in shape)).astype(np.float64)
        b = np.roll(a, (2, -3), axis=(0, 1)) + rng.normal(size=a.shape)
        m = g["maxabs"][k]
        below, above = np.nonzero(m < lo)[0], np.nonzero(m > hi)[0]
        for j in (below[-1], below[len(below) // 2], above[0], above[-1]):      # next to the band on either side, and far from it
            s_ = float(g["scales"][j])
            got = gs.xcorr2d(a * s_, b * s_, normalize="none")[0]
            assert np.iscomplexobj(got) == bool(g["is_complex"][k, j]), (tuple(shape), float(m[j]))
            want = S.xcorr2d(a * s_, b * s_, normalize="none")[0]
            assert int(np.argmax(got)) == int(np.argmax(want))
            checked += 1
    assert checked == 4 * len(g["shapes"])


def test_xcorr2d_detector_counts_are_complex128(gs):
    """The normal case: un-standardised detector data -> complex128, like the reference (oracle)."""
    from oracle import signal_np as S

    a = synth.speckle_frame(256, 3)
    b = np.roll(a, (3, -5), axis=(0, 1))
    got = gs.xcorr2d(a, b)[0]
    want = S.xcorr2d(a.astype(np.float64), b.astype(np.float64))[0]
    assert got.dtype == want.dtype == np.complex128
    assert int(np.argmax(got)) == int(np.argmax(want)) and np.all(got.imag == 0)
    assert nerr(got.real, want.real) < TOL


def test_thread_reentrancy(gs):
    """The reference's stack functions call the per-frame entry points from joblib threads (speckles.py:323): the
    C ABI must tolerate several host threads on one plan / the shared scratch.  8 threads x mixed calls == serial."""
    from concurrent.futures import ThreadPoolExecutor

    from barc4dip_amd import metrics as gm
    from barc4dip_amd import synth

    frames = [synth.speckle_frame(512, 300 + i) for i in range(8)]

    def work(f):
        return (gs.psd2d(f)[0], gs.autocorr2d(f)[0], gm.distribution_moments(f)["kurtosis"], gm.speckles.amplitude(f)["contrast"],
                gs.phase_correlation(f[100:221, 100:221], f, slices_yx=(slice(100, 221), slice(100, 221)))[:2])

    serial = [work(f) for f in frames]
    with ThreadPoolExecutor(max_workers=8) as ex:
        threaded = list(ex.map(work, frames))
    for a, b in zip(serial, threaded):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and a[4] == b[4]


@pytest.mark.parametrize("shape", [(64, 64), (100, 37), (512, 512), (300, 520), (1024, 2048)])
def test_complex_fft2d_and_ifft2d(gs, shape):
    """Complex frames (signal/fft.py:198-258): fft2d through the complex-to-complex engine, ifft2d on the device,
    round trip ifft2d(fft2d(x)) == x; complex64 in -> complex64 out, complex128 in -> complex128 out."""
    from oracle import signal_np as S

    rng = np.random.default_rng(shape[0] + shape[1])
    z = (rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(np.complex64)
    F, fx, fy = gs.fft2d(z, dx=0.5)
    Fr, fxr, fyr = S.fft2d(z.astype(np.complex128), dx=0.5)
    assert F.dtype == np.complex64 and nerr(F, Fr) < TOL and np.array_equal(fx, fxr) and np.array_equal(fy, fyr)
    back = gs.fft.ifft2d(F)
    assert back.dtype == np.complex64 and nerr(back, z) < TOL
    assert nerr(gs.fft.ifft2d(Fr), S.ifft2d(Fr)) < TOL and gs.fft.ifft2d(Fr).dtype == np.complex128
    real = rng.random(shape).astype(np.float32)
    assert nerr(gs.fft.ifft2d(gs.fft2d(real)[0]).real, real) < TOL


@pytest.mark.parametrize("shape", [(1072, 536), (1197, 640), (1540, 1326), (1326, 1072), (3000, 520), (8192, 64)])
def test_fused_transform_factorisations(gs, shape):
    """Sides chosen for the corner cases of the fused row transform: a prime M (1072 = 16 * 67: A = 67, B = 1, table-free
    path), P = 1 with A > 32 (1197 = 63 * 19), three-level splits (1540 = 4 * 35 * 11, 1326 = 2 * 39 * 17), the one-buffer
    variant (536, 640, 520) next to the two-buffer one, and the longest supported row (8192)."""
    from oracle import signal_np as S

    rng = np.random.default_rng(shape[0] ^ shape[1])
    img = (rng.random(shape) * 100).astype(np.float32)
    r64 = img.astype(np.float64)
    assert nerr(gs.fft2d(img)[0], S.fft2d(r64)[0]) < TOL
    ac = gs.autocorr2d(img)[0]
    assert nerr(ac, S.autocorr2d(r64)[0]) < TOL and ac[shape[0] // 2, shape[1] // 2] == 1.0


@pytest.mark.parametrize("shape", [(1042, 1042), (2056, 2464), (1031, 520), (600, 4093)])
def test_bluestein_lengths_vs_oracle(gs, shape):
    """Sides with a large prime factor (1042 = 2 * 521, 2056 = 8 * 257, 1031 and 4093 prime) have no small-factor
    split: chirp-z over two fused power-of-two transforms.  Same 1e-5 bar."""
    from oracle import signal_np as S

    rng = np.random.default_rng(shape[0] + 3 * shape[1])
    img = (rng.poisson(300.0, size=shape) + rng.random(shape)).astype(np.float32)
    r64 = img.astype(np.float64)
    assert nerr(gs.fft2d(img)[0], S.fft2d(r64)[0]) < TOL
    assert nerr(gs.psd2d(img)[0], S.psd2d(r64)[0]) < TOL
    ac = gs.autocorr2d(img)[0]
    assert nerr(ac, S.autocorr2d(r64)[0]) < TOL and ac[shape[0] // 2, shape[1] // 2] == 1.0
    z = (rng.normal(size=shape) + 1j * rng.normal(size=shape)).astype(np.complex64)
    assert nerr(gs.fft.ifft2d(gs.fft2d(z)[0]), z) < TOL


def test_library_loaded_before_torch_still_sees_the_gpu():
    """Touching the C ABI (introspection) before anything imported torch must not strand libb4d.so on a second,
    device-less copy of the HIP runtime (_ffi.load_library pulls torch's runtime in first)."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, '.'); from barc4dip_amd import _ffi; assert _ffi.supported(64, 64); "
            "import numpy as np; from barc4dip_amd import signal as s; "
            "p = s.psd2d(np.arange(4096, dtype=np.float32).reshape(64, 64))[0]; print(p.shape, float(p.max()) > 0)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "(64, 64) True" in out.stdout, out.stderr[-500:]


@pytest.mark.parametrize("shape", [(64, 4096), (4096, 64), (128, 1024), (4096, 512), (2048, 256), (256, 2048), (1024, 1024)])
def test_fft2d_columns_first_rows_last(gs, shape):
    """fft2d of real frames at power-of-two sizes (b4d_spectrum.hip: column pass on real columns, row pass writes every output row
    and its conjugate mirror): both column orders of the intermediate (32- and 16-column tiles, ny = 4096), extreme aspect ratios,
    a stack longer than the plan's launch group, and the symmetries the packing relies on.  signal/fft.py:198-237."""
    from barc4dip_amd.signal.fft import fft2d_stack
    from oracle import signal_np as S

    ny, nx = shape
    T = 67 if ny * nx <= 1 << 18 else 3      # 67 > the default launch group of 64 frames
    rng = np.random.default_rng(ny * 7 + nx)
    st = (rng.poisson(200.0, size=(T, ny, nx)) + rng.normal(size=(T, ny, nx))).astype(np.float32)
    F = fft2d_stack(st)
    assert F.shape == (T, ny, nx) and F.dtype == np.complex64
    for t in (0, T - 1):
        Fr = S.fft2d(st[t].astype(np.float64))[0]
        assert nerr(F[t], Fr) < TOL
        e32 = nerr(S.fft2d(st[t])[0], Fr)
        assert nerr(F[t], Fr) <= max(4 * e32, 2e-7)
    # Hermitian symmetry of a real frame's spectrum, exactly as stored: F[-ky, -kx] = conj F[ky, kx] bit for bit (the mirror row
    # is the SAME value conjugated), the four self-conjugate bins are real
    G = F[T // 2]
    M = np.roll(G[::-1, ::-1], (1, 1), axis=(0, 1))
    inner = (slice(1, None), slice(1, None))
    np.testing.assert_array_equal(G[inner], np.conj(M)[inner])
    for (y, x) in ((ny // 2, nx // 2), (0, nx // 2), (ny // 2, 0), (0, 0)):
        assert G[y, x].imag == 0.0


def test_lanes_option_is_only_a_route(gs):
    """b4d_set_option("lanes", 0 / 1) (include/b4d.h): the two-lane launch groups (cache-sized groups dealt to the caller's stream
    and the library's second one) against everything on the caller's stream: bit-identical outputs for fft2d (power-of-two and
    mixed-radix sizes), psd + autocorr at a detector format, the Wiener mixed-radix driver and phase-correlation pair groups, at
    stack lengths that do not divide into the groups."""
    import torch

    from barc4dip_amd import _ffi
    from barc4dip_amd.preprocessing import deconvolve_psf
    from barc4dip_amd.signal.fft import fft2d_stack

    lib = _ffi.lib()
    rng = np.random.default_rng(5)
    pow2 = torch.from_numpy((rng.poisson(300.0, size=(41, 1024, 1024))).astype(np.float32)).cuda()
    det = torch.from_numpy((rng.poisson(300.0, size=(13, 1080, 1920))).astype(np.float32)).cuda()
    small = torch.from_numpy((rng.poisson(300.0, size=(7, 512, 512))).astype(np.float32)).cuda()
    stack, _ = synth.shifted_stack(4, 256, seed=3, max_shift=6)
    rois = [(y, y + 61, x, x + 61) for y in (20, 90, 160) for x in (25, 95, 165)]
    tpl_frame = [0] * 9 + [max(t - 1, 0) for t in range(4) for _ in range(9)]
    tpl_roi = rois + rois * 4
    pair_img = [t for t in range(4) for _ in range(9)] * 2
    pair_tpl = [k for _ in range(4) for k in range(9)] + [9 + 9 * t + k for t in range(4) for k in range(9)]
    out = {}
    try:
        for lanes in (1, 0):
            assert lib.b4d_set_option(b"lanes", lanes) == 0
            psd, ac = gs.psd_autocorr2d_stack(det, return_tensors=True)[:2]
            out[lanes] = (fft2d_stack(pow2, return_tensors=True).cpu().numpy(), fft2d_stack(det, return_tensors=True).cpu().numpy(),
                          psd.cpu().numpy(), ac.cpu().numpy(), deconvolve_psf(small, sigma=1.5, return_tensors=True).cpu().numpy(),
                          gs.phase_correlation_batch(stack, stack, tpl_frame, tpl_roi, pair_img, pair_tpl))
            torch.cuda.synchronize()
    finally:
        lib.b4d_set_option(b"lanes", 1)
    assert lib.b4d_set_option(b"lanes", 2) != 0
    for a, b in zip(out[1], out[0]):
        np.testing.assert_array_equal(a, b)
    # the power-of-two psd + autocorr pipeline takes two lanes from 4096^2 frames on
    big = torch.from_numpy(rng.poisson(300.0, size=(5, 4096, 4096)).astype(np.float32)).cuda()
    res = {}
    try:
        for lanes in (1, 0):
            assert lib.b4d_set_option(b"lanes", lanes) == 0
            psd, ac = gs.psd_autocorr2d_stack(big, return_tensors=True)[:2]
            res[lanes] = (psd.cpu().numpy(), ac.cpu().numpy())
    finally:
        lib.b4d_set_option(b"lanes", 1)
    np.testing.assert_array_equal(res[1][0], res[0][0])
    np.testing.assert_array_equal(res[1][1], res[0][1])
    assert res[1][1][4, 2048, 2048] == 1.0
    del big, res
    # b4d_xcorr2d over a stack of pairs (the Python signature takes one pair): power-of-two and mixed-radix plans
    import ctypes as C

    for frames in (pow2[:24], det[:12]):
        T, ny, nx = (int(v) for v in frames.shape)
        plan = _ffi.get_plan(ny, nx, general=not (ny & (ny - 1) == 0 and nx & (nx - 1) == 0))
        other = torch.roll(frames, shifts=(3, -5), dims=(1, 2)).contiguous()
        res = {}
        try:
            for lanes in (1, 0):
                assert lib.b4d_set_option(b"lanes", lanes) == 0
                corr = torch.empty_like(frames)
                _ffi.check(lib.b4d_xcorr2d(plan.handle, C.c_void_p(frames.data_ptr()), C.c_void_p(other.data_ptr()), T,
                                           C.c_void_p(corr.data_ptr()), _ffi.REMOVE_MEAN | _ffi.NORM_PEAK, _ffi.stream_ptr()))
                res[lanes] = corr.cpu().numpy()
        finally:
            lib.b4d_set_option(b"lanes", 1)
        np.testing.assert_array_equal(res[1], res[0])
        peak = np.unravel_index(int(np.argmax(res[1][T - 1])), (ny, nx))
        assert peak == (ny // 2 - 3, nx // 2 + 5), peak
    from oracle import signal_np as S

    assert nerr(out[1][0][40], S.fft2d(pow2[40].cpu().numpy().astype(np.float64))[0]) < TOL
    assert nerr(out[1][1][12], S.fft2d(det[12].cpu().numpy().astype(np.float64))[0]) < TOL
