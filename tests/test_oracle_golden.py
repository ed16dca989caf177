"""CPU tier: the NumPy oracle reproduces the vectors captured from the real reference
(tests/golden/*.npz, written by oracle/make_golden.py).  Same NumPy calls => bit-exact for
the signal layer; metrics scalars to 1e-12 relative (BLAS/threading order only)."""
import warnings

import numpy as np
import pytest

from barc4dip_amd import synth
from oracle import metrics_np as M
from oracle import signal_np as S

CASES = ["f64_24x32", "f32_32x16", "f64_17x23", "f32_64"]


@pytest.mark.parametrize("name", CASES)
def test_fft_psd_bit_exact(golden, name):
    g = golden("signal_small.npz")
    a = g[f"{name}/a"]
    F, fx, fy = S.fft2d(a, dx=0.5, dy=2.0)
    assert F.dtype == g[f"{name}/fft2d"].dtype
    np.testing.assert_array_equal(F, g[f"{name}/fft2d"])
    np.testing.assert_array_equal(fx, g[f"{name}/fx"])
    np.testing.assert_array_equal(fy, g[f"{name}/fy"])
    np.testing.assert_array_equal(S.ifft2d(F), g[f"{name}/ifft2d"])
    for key, kw in (("psd2d", {}), ("psd2d_cal", dict(dx=0.5, dy=2.0)), ("psd2d_noscale", dict(scale=False))):
        P = S.psd2d(a, **kw)[0]
        assert P.dtype == g[f"{name}/{key}"].dtype
        np.testing.assert_array_equal(P, g[f"{name}/{key}"])


@pytest.mark.parametrize("name", CASES)
def test_xcorr_autocorr_bit_exact(golden, name):
    g = golden("signal_small.npz")
    a, b = g[f"{name}/a"], g[f"{name}/b"]
    n = 0
    for rm in (True, False):
        for st in (True, False):
            for nm in ("peak", "none"):
                tag = f"rm{int(rm)}_st{int(st)}_{nm}"
                if f"{name}/xcorr2d_{tag}" not in g:
                    continue
                n += 1
                c, xl, yl = S.xcorr2d(a, b, remove_mean=rm, standardize=st, normalize=nm)
                assert c.dtype == g[f"{name}/xcorr2d_{tag}"].dtype
                np.testing.assert_array_equal(c, g[f"{name}/xcorr2d_{tag}"])
                ac = S.autocorr2d(a, remove_mean=rm, standardize=st, normalize=nm)[0]
                np.testing.assert_array_equal(ac, g[f"{name}/autocorr2d_{tag}"])
    assert n >= 2
    np.testing.assert_array_equal(xl, g[f"{name}/xlag"])
    np.testing.assert_array_equal(yl, g[f"{name}/ylag"])


def _xcorr_dtype_inputs(shape, seed):
    """The seeded pair of oracle/make_golden_xcorr_dtype.py."""
    rng = np.random.default_rng(int(seed))
    a = rng.poisson(50.0, size=tuple(int(v) for v in shape)).astype(np.float64)
    b = np.roll(a, (2, -3), axis=(0, 1)) + rng.normal(size=a.shape)
    return a, b


def test_xcorr2d_return_type_sweep(golden):
    """np.real_if_close decides xcorr2d's dtype on rounding noise (signal/corr.py:41-42, 242): the oracle performs the reference's
    arithmetic, so it reproduces the reference's dtype at EVERY amplitude of the sweep -- inside the data-dependent band as well."""
    g = golden("xcorr_dtype.npz")
    lo, hi = g["band"]
    assert 1000 < lo < 4000 < hi < 20000        # the product's rule (complex128 beyond max|corr| = 4000) sits inside the band
    for k, (shape, seed) in enumerate(zip(g["shapes"], g["seeds"])):
        a, b = _xcorr_dtype_inputs(shape, seed)
        for j in range(0, len(g["scales"]), 5):
            c = S.xcorr2d(a * g["scales"][j], b * g["scales"][j], normalize="none")[0]
            assert np.iscomplexobj(c) == bool(g["is_complex"][k, j]), (tuple(shape), j)
            assert float(np.max(np.abs(c))) == pytest.approx(float(g["maxabs"][k, j]), rel=1e-12)


def test_1d_helpers(golden):
    g = golden("signal_small.npz")
    a, b = g["1d/a"], g["1d/b"]
    F, fx = S.fft1d(a, dx=0.25)
    np.testing.assert_array_equal(F, g["1d/fft1d"])
    np.testing.assert_array_equal(fx, g["1d/fx"])
    np.testing.assert_array_equal(S.ifft1d(F), g["1d/ifft1d"])
    np.testing.assert_array_equal(S.psd1d(a, dx=0.25)[0], g["1d/psd1d"])
    np.testing.assert_array_equal(S.psd1d(a, scale=False)[0], g["1d/psd1d_noscale"])
    c, lag = S.xcorr1d(a, b)
    np.testing.assert_array_equal(c, g["1d/xcorr1d"])
    np.testing.assert_array_equal(lag, g["1d/lag"])
    np.testing.assert_array_equal(S.xcorr1d(a, a, standardize=True)[0], g["1d/autocorr1d"])
    np.testing.assert_array_equal(S.freq_axis1d(n=100, x=g["1d/xs"]), g["1d/freq_axis_x"])


def test_axis_validation():
    with pytest.raises(ValueError):
        S.fft2d(np.zeros(4))
    with pytest.raises(ValueError):
        S.psd2d(np.zeros((4, 4)), x=np.arange(4.0))
    with pytest.raises(ValueError):
        S.psd2d(np.zeros((4, 4)), x=np.arange(4.0), y=np.arange(4.0), dx=2.0)
    with pytest.raises(ValueError):
        S.fft2d(np.zeros((4, 4)), x=np.array([0, 1, 2, 4.0]), y=np.arange(4.0))
    with pytest.raises(ValueError):
        S.xcorr2d(np.zeros((4, 4)), np.zeros((4, 5)))
    with pytest.raises(ValueError):
        S.xcorr2d(np.zeros((4, 4)), np.zeros((4, 4)), normalize="bogus")
    with pytest.raises(ValueError):
        S.roi_slices((10, 10), (4, 5))       # even size
    with pytest.raises(ValueError):
        S.roi_slices((10, 10), (11, 5))      # out of bounds


def _phase_inputs():
    i0 = synth.speckle_intensity(256, 1234)
    f0 = np.random.default_rng(1).poisson(i0).astype(np.float32)
    return i0, f0


def test_phase_correlation_rows(golden):
    g = golden("tracking.npz")
    i0, f0 = _phase_inputs()
    rows = g["phase/rows"]
    frames = {}
    for r in rows:
        k, sy, sx, side, cy, cx, sub, bits = (int(v) for v in r[:8])
        if k not in frames:
            frames[k] = np.random.default_rng(100 + k).poisson(np.roll(i0, (sy, sx), axis=(0, 1))).astype(np.float32)
        dt = np.float32 if bits == 32 else np.float64
        sl = S.roi_slices((256, 256), (side, side), center_yx=None if cy < 0 else (cy, cx))
        got = S.phase_correlation(f0[sl].astype(dt), frames[k].astype(dt), slices_yx=sl, subpixel=bool(sub))
        np.testing.assert_array_equal(np.asarray(got, dtype=np.float64), r[8:])
        if side >= 121:           # well-conditioned: integer part equals the imposed shift
            assert round(got[0]) == sy and round(got[1]) == sx


def test_track_dispatch_and_taylor(golden):
    g = golden("tracking.npz")
    i0, f0 = _phase_inputs()
    fr = np.random.default_rng(101).poisson(np.roll(i0, (3, -5), axis=(0, 1))).astype(np.float32)
    sl = S.roi_slices((256, 256), (121, 121))
    np.testing.assert_array_equal(np.asarray(S.track_translation(f0[sl], fr)), g["track/default"])
    out = np.asarray([S.peak_subpixel_taylor(m, (2, 2)) for m in g["taylor/in"]])
    np.testing.assert_array_equal(out, g["taylor/out"])
    assert S.peak_subpixel_taylor(g["taylor/in"][0], (0, 2)) == (0.0, 0.0)
    with pytest.raises(ValueError):
        S.track_translation(f0[sl], fr, method="bogus")
    with pytest.raises(ValueError):
        S.track_translation(f0[sl], fr, method="template")      # default backend "internal" is invalid there
    with pytest.raises(ValueError):
        S.phase_correlation(f0[:120, :120], fr)                  # even template, slices_yx=None
    got = S.phase_correlation(g["map64/f0"][S.roi_slices((64, 64), (31, 31))], g["map64/f1"],
                              slices_yx=S.roi_slices((64, 64), (31, 31)))
    np.testing.assert_array_equal(np.asarray(got), g["map64/result"])


def _walk(prefix, d, g, rtol, seen):
    for k, v in d.items():
        key = f"{prefix}/{k}"
        if isinstance(v, dict):
            _walk(key, v, g, rtol, seen)
        elif key in g.files and g[key].dtype.kind not in "US":
            np.testing.assert_allclose(np.asarray(v, dtype=float), g[key], rtol=rtol, atol=0, equal_nan=True,
                                       err_msg=key)
            seen.append(key)


def _kat_image(n=512):
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[-n // 2:n // 2, -n // 2:n // 2]
    pupil = (xx ** 2 + yy ** 2) <= (n / 16) ** 2
    field = np.fft.ifft2(np.fft.ifftshift(pupil * np.exp(2j * np.pi * rng.random((n, n)))))
    img = np.abs(field) ** 2
    return (img / img.mean() * 1000).astype(np.float32)


@pytest.mark.parametrize("tag", ["kat512", "poisson512"])
@pytest.mark.parametrize("origin", ["lower", "upper"])
def test_aggregators_512(golden, tag, origin):
    g = golden("metrics.npz")
    img = _kat_image() if tag == "kat512" else synth.speckle_frame(512, 1234)
    seen = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sp = M.speckle_stats(img, display_origin=origin)
        sh = M.sharpness_stats(img, display_origin=origin)
    ac = sp["full"]["grain"].pop("autocorr")
    np.testing.assert_allclose(ac[256, :], g[f"{tag}/{origin}/speckle/full/grain/autocorr_cut_x"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(ac[:, 256], g[f"{tag}/{origin}/speckle/full/grain/autocorr_cut_y"], rtol=1e-12, atol=1e-15)
    _walk(f"{tag}/{origin}/speckle", sp, g, 1e-10, seen)
    _walk(f"{tag}/{origin}/sharpness", sh, g, 1e-10, seen)
    assert len(seen) > 60
    assert sp["meta"]["tile_mode"] == "tiles_3x3"


def test_survey_kats():
    """Known answers recorded in SURVEY.md §8c from the reference (independent of our fixtures)."""
    img = _kat_image()
    a = M.amplitude(img)
    assert a["visibility"] == pytest.approx(0.9955749775726361, rel=1e-12)
    assert a["contrast"] == pytest.approx(0.9998632364734213, rel=1e-12)
    gr = M.grain(img)
    assert gr["lx"] == pytest.approx(9.690115608100314, rel=1e-10)
    assert gr["ly"] == pytest.approx(9.660333123945321, rel=1e-10)
    assert gr["leq"] == pytest.approx(9.663208059536487, rel=1e-10)
    bw = M.bandwidth(img)
    assert bw["feq"] == pytest.approx(0.06289603897556885, rel=1e-10)
    assert bw["f95"] == pytest.approx(0.10150614422020533, rel=1e-12)
    assert bw["spr"] == pytest.approx(3621.6068899820348, rel=1e-10)
    assert M.tenengrad(img)["tenengrad"] == pytest.approx(8973629.89178935, rel=1e-10)
    assert M.laplacian_variance(img) == pytest.approx(38699.84234914002, rel=1e-10)
    assert M.spectral_entropy(img) == pytest.approx(0.6911419299004048, rel=1e-10)
    e = M.eigenvalues(img)
    assert e["eigenvalues"] == pytest.approx(4.2286742100154455e-07, rel=1e-9)
    dm = M.distribution_moments(img)
    assert dm["skewness"] == pytest.approx(2.0379286254969116, rel=1e-10)
    assert dm["kurtosis"] == pytest.approx(6.426728078998179, rel=1e-10)
    ac = S.autocorr2d(img)[0]
    assert np.unravel_index(np.argmax(ac), ac.shape) == (256, 256) and ac[256, 256] == 1.0
    r = S.phase_correlation(img, np.roll(img, (3, -5), axis=(0, 1)), slices_yx=(slice(0, 512), slice(0, 512)))
    assert r[0] == pytest.approx(2.9996219049207866, rel=1e-6)
    assert r[1] == pytest.approx(-4.999248978798278, rel=1e-6)


def test_single_metrics_odd_shape(golden):
    g = golden("metrics.npz")
    n, seed, h, w = (int(v) for v in g["odd/in_seed"])
    odd = synth.speckle_frame(n, seed)[:h, :w].astype(np.float64)
    seen = []
    gr = M.grain(odd)
    _walk("odd", {"amplitude": M.amplitude(odd),
                  "grain": {k: gr[k] for k in ("lx", "ly", "leq", "r")},
                  "grain_binned": {k: M.grain(odd, radial_method="binned")[k] for k in ("lx", "ly", "leq", "r")},
                  "bandwidth": M.bandwidth(odd), "tenengrad": M.tenengrad(odd),
                  "laplacian_variance": M.laplacian_variance(odd),
                  "spectral_entropy": M.spectral_entropy(odd),
                  "inverse_autocorr_width": M.inverse_autocorr_width(odd),
                  "eigenvalues": M.eigenvalues(odd),
                  "moments": M.distribution_moments(odd, saturation_value=3000.0)}, g, 1e-10, seen)
    assert len(seen) >= 30
    bad = odd.copy()
    bad[5, 7] = np.nan
    bad[100, 3] = np.inf
    _walk("nan", {"moments": M.distribution_moments(bad),
                  "tenengrad": M.tenengrad(np.where(np.isfinite(bad), bad, 0.0))}, g, 1e-12, seen)
    with pytest.raises(ValueError):
        M.spectral_entropy(bad)
    with pytest.raises(ValueError):
        M.grain(odd[:100, :100])
    with pytest.raises(ValueError):
        M.amplitude(-odd)


def test_maths_helpers(golden):
    g = golden("metrics.npz")
    prof = g["maths/profile"]
    assert M.width_at_fraction(prof)[0] == g["maths/width"]
    assert M.width_at_fraction(prof, fraction=0.5, center_index=50)[0] == g["maths/width_half"]
    assert M.distance_at_fraction_from_peak(prof[50:], fraction=0.5)[0] == g["maths/dist"]
    assert M.width_at_fraction(np.ones(10) + np.arange(10)) == (float(g["maths/width_edge"]), True)
    acs = S.autocorr2d(synth.speckle_frame(128, 5))[0]
    np.testing.assert_allclose(M.radial_mean_interpolated(acs)[0], g["maths/radial_interp"], rtol=1e-12)
    np.testing.assert_allclose(M.radial_mean_binned(acs)[0], g["maths/radial_binned"], rtol=1e-12)


def test_stack_stats(golden):
    g = golden("stack.npz")
    stack, sh = synth.shifted_stack(5, 384, seed=1234, max_shift=12)
    np.testing.assert_array_equal(sh, g["shifts"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = M.speckle_stack_stats(stack, metrics=("amplitude", "grain", "stats"), tiles=True,
                                    roi_grain_factor=24.0, tracking_method="phase", tracking_backend="internal")
        res2 = M.sharpness_stack_stats(stack[:3], metrics=("gradient", "laplacian", "spectral"))
    res["full"]["grain"].pop("autocorr")
    seen = []
    _walk("speckle", {k: res[k] for k in ("full", "tiles", "temporal")}, g, 1e-9, seen)
    _walk("sharpness", {k: res2[k] for k in ("full", "tiles")}, g, 1e-9, seen)
    assert len(seen) > 40
    np.testing.assert_array_equal(np.asarray(res["meta"]["tracking"]["roi_size_yx"]), g["speckle/meta/roi_size_yx"])
    # ground truth: the tracker recovers the imposed integer shifts on every ROI
    np.testing.assert_allclose(res["temporal"]["abs"]["dy"], sh[:, 0], atol=0.2)
    np.testing.assert_allclose(res["temporal"]["abs"]["dx"], sh[:, 1], atol=0.2)
    with pytest.raises(TypeError):
        M.speckle_stack_stats([1, 2, 3])
    with pytest.raises(ValueError):
        M.speckle_stack_stats(stack[0])
