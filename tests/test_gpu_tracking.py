"""GPU tier: phase-correlation tracking and xcorr2d through the C ABI vs the oracle and the
golden vectors captured from the reference (tests/golden/tracking.npz).

Bar: integer translation (arg-max) bit-exact.  Peak, SNR and the sub-pixel part come from a
WHITENED spectrum (every bin scaled to unit modulus, so bins whose cross-power is rounding noise
carry random phases): the reference's own float32 and float64 paths differ by ~2e-4 relative in
the peak and ~3e-4 in SNR on these inputs (see the golden rows).  We therefore compare with the
reference's FLOAT64 rows with bars held at 2 x the deviation observed on MI355X (BARS below: e.g. 256^2 golden rows 9e-4
peak, 2.2e-3 SNR, 6e-5 px sub-pixel part), and check that we are no further from float64 than 4x the reference's own
float32 path."""
import numpy as np
import pytest

from barc4dip_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gs():
    import torch

    assert torch.cuda.is_available()
    from barc4dip_amd import signal

    return signal


# Tolerance bars of the tests below = 2 x the largest deviation observed on MI355X (round 3, tests/conftest.py::observe writes the
# maxima of every run to gpurun_out/observed_tolerances.json; DESIGN.md section 5 holds the table).  Keys: <test>/<quantity>.
BARS = {   # observed maximum in the comment
    "golden_rows_256/sub_px": 6e-5,    # 2.7e-5 px   (256^2, 36 golden rows of the reference, float64 protocol)
    "golden_rows_256/peak_rel": 9e-4,  # 4.4e-4
    "golden_rows_256/snr_rel": 2.2e-3,  # 1.07e-3    (61-px templates: the whitened spectrum leaves rounding noise undamped)
    "batch_512/sub_px": 1.4e-4,        # 6.6e-5 px   (512^2, 90 pairs)
    "batch_512/peak_rel": 5.5e-4,      # 2.6e-4
    "batch_512/snr_rel": 1.5e-3,       # 7.2e-4
    "general_sizes/sub_px": 1.4e-4,    # 6.7e-5 px   (200x300 ... 1080x1920: DFT-matrix and mixed-radix routes)
    "general_sizes/peak_rel": 6.5e-4,  # 3.0e-4
    "general_sizes/snr_rel": 1.3e-3,   # 6.2e-4
    "all_routes/sub_px": 3e-5,         # 1.5e-5 px   (64^2 ... 4096x1024, three median routes)
    "all_routes/peak_rel": 1.2e-3,     # 5.6e-4
    "all_routes/snr_rel": 1.3e-3,      # 6.1e-4
    # NCC template matching: float32 rounding level (observed 1.2e-7 ... 4.8e-7); the floor is held at 2e-6 rather than 2 x that
    "ncc_256/peak_abs": 2e-6, "ncc_256/sub_px": 2e-6, "ncc_256/snr_rel": 2e-6,
    "ncc_batch/peak_abs": 2e-6,
    "ncc_mixed/sub_px": 2e-6, "ncc_mixed/peak_abs": 2e-6, "ncc_mixed/snr_rel": 2e-6,
}


def _rel(got, want):
    return abs(got - want) / abs(want)


def _inputs():
    i0 = synth.speckle_intensity(256, 1234)
    f0 = np.random.default_rng(1).poisson(i0).astype(np.float32)
    return i0, f0


def test_golden_rows_256(gs, golden, observe):
    from barc4dip_amd.geometry import roi_slices

    g = golden("tracking.npz")
    i0, f0 = _inputs()
    rows = g["phase/rows"]
    frames = {}
    checked = 0
    ref32 = {tuple(int(v) for v in r[:7]): r[8:] for r in rows if int(r[7]) == 32}
    for r in rows:
        k, sy, sx, side, cy, cx, sub, bits = (int(v) for v in r[:8])
        if bits != 64:
            continue
        if k not in frames:
            frames[k] = np.random.default_rng(100 + k).poisson(np.roll(i0, (sy, sx), axis=(0, 1))).astype(np.float32)
        sl = roi_slices((256, 256), (side, side), center_yx=None if cy < 0 else (cy, cx))
        dy, dx, peak, snr = gs.phase_correlation(f0[sl], frames[k], slices_yx=sl, subpixel=bool(sub))
        rdy, rdx, rpeak, rsnr = r[8:]
        assert round(dy) == round(rdy) and round(dx) == round(rdx), (r[:8], dy, dx)
        if not sub:
            assert dy == rdy and dx == rdx                      # integer outputs: bit exact
        else:
            observe("golden_rows_256/sub_px", max(abs(dy - rdy), abs(dx - rdx)), BARS["golden_rows_256/sub_px"])
        observe("golden_rows_256/peak_rel", _rel(peak, rpeak), BARS["golden_rows_256/peak_rel"])
        observe("golden_rows_256/snr_rel", _rel(snr, rsnr), BARS["golden_rows_256/snr_rel"])
        own = ref32[(k, sy, sx, side, cy, cx, sub)]            # the reference's float32 path on the same input
        assert abs(peak - rpeak) <= 4 * abs(own[2] - rpeak) + 1e-6 * rpeak
        checked += 1
    assert checked == 36


def test_dispatcher_errors_and_defaults(gs, golden):
    from barc4dip_amd.geometry import roi_slices

    g = golden("tracking.npz")
    i0, f0 = _inputs()
    fr = np.random.default_rng(101).poisson(np.roll(i0, (3, -5), axis=(0, 1))).astype(np.float32)
    sl = roi_slices((256, 256), (121, 121))
    got = np.asarray(gs.track_translation(f0[sl], fr))
    np.testing.assert_allclose(got, g["track/default"], rtol=2e-3, atol=5e-3)
    assert round(got[0]) == 3 and round(got[1]) == -5
    with pytest.raises(ValueError):
        gs.track_translation(f0[sl], fr, method="bogus")
    with pytest.raises(ValueError):
        gs.track_translation(f0[sl], fr, method="template")            # backend "internal" invalid there
    r_t = gs.track_translation(f0[sl], fr, method="template", backend="skimage")   # NCC back-end runs on the device
    assert round(r_t[0]) == 3 and round(r_t[1]) == -5
    from oracle import phase_skimage_np as P

    r_s = gs.phase_correlation(f0[sl], fr, backend="skimage")           # built from the published algorithm: no ImportError
    w_s = P.phase_correlation_skimage(f0[sl], fr)
    assert abs(r_s[0] - w_s[0]) < 1e-9 and abs(r_s[1] - w_s[1]) < 1e-9 and np.isnan(r_s[2]) and np.isnan(r_s[3])
    assert round(r_s[0]) == 3 and round(r_s[1]) == -5                   # (shot noise: the 0.1-px estimate itself is 2.8, -4.8 ...)
    with pytest.raises(ValueError):
        gs.phase_correlation(f0[:120, :120], fr)                        # even template without slices
    with pytest.raises(ValueError):
        gs.phase_correlation(f0[sl], fr, slices_yx=(slice(0, 100), slice(0, 100)))
    # 64x64 full-map golden case (smallest native size)
    s64 = roi_slices((64, 64), (31, 31))
    r = gs.phase_correlation(g["map64/f0"][s64], g["map64/f1"], slices_yx=s64)
    ref = g["map64/result"]
    assert round(r[0]) == round(ref[0]) == 2 and round(r[1]) == round(ref[1]) == -3


def test_batch_matches_oracle_and_truth(gs, observe):
    """cfg3 protocol at reduced T: 3x3 ROI grid, abs + inc templates, all shifts recovered integer-exact."""
    from barc4dip_amd.geometry import roi_grid_3x3
    from oracle import signal_np as S

    T, n = 5, 512
    stack, sh = synth.shifted_stack(T, n, seed=1234, max_shift=24)
    grid, _ = roi_grid_3x3((n, n), (121, 121), (60, 60))
    rois = [(s[0].start, s[0].stop, s[1].start, s[1].stop) for s in grid.ravel()]
    tpl_frame, tpl_roi, pair_img, pair_tpl = [], [], [], []
    for t in range(T):
        for k, r in enumerate(rois):       # abs: frame-0 templates, shared by every t
            if t == 0:
                tpl_frame.append(0)
                tpl_roi.append(r)
            pair_img.append(t)
            pair_tpl.append(k)
    for t in range(T):                      # inc: templates from the previous frame (t = 0 uses frame 0)
        for k, r in enumerate(rois):
            tpl_frame.append(max(t - 1, 0))
            tpl_roi.append(r)
            pair_img.append(t)
            pair_tpl.append(len(tpl_frame) - 1)
    res, pij = gs.phase_correlation_batch(stack, stack, tpl_frame, tpl_roi, pair_img, pair_tpl, return_peak_ij=True)
    npairs = len(pair_img)
    assert res.shape == (npairs, 4)
    for i in range(npairs):                 # EVERY pair against the oracle (~50 ms each on the CPU)
        r = rois[i % 9]
        sl = (slice(r[0], r[1]), slice(r[2], r[3]))
        ref = S.phase_correlation(stack[tpl_frame[pair_tpl[i]]][sl], stack[pair_img[i]], slices_yx=sl)
        mag = S.phase_correlation_map(stack[tpl_frame[pair_tpl[i]]][sl], stack[pair_img[i]], slices_yx=sl)
        mi, mj = np.unravel_index(np.argmax(mag), mag.shape)
        assert (pij[i, 0], pij[i, 1]) == (mi, mj)                      # index output: bit exact
        observe("batch_512/sub_px", max(abs(res[i, 0] - ref[0]), abs(res[i, 1] - ref[1])), BARS["batch_512/sub_px"])
        observe("batch_512/peak_rel", _rel(res[i, 2], ref[2]), BARS["batch_512/peak_rel"])
        observe("batch_512/snr_rel", _rel(res[i, 3], ref[3]), BARS["batch_512/snr_rel"])
    # ground truth: abs shifts equal the imposed spiral, inc shifts its differences
    abs_dy = res[:T * 9, 0].reshape(T, 9)
    abs_dx = res[:T * 9, 1].reshape(T, 9)
    assert np.all(np.rint(abs_dy) == sh[:, 0:1]) and np.all(np.rint(abs_dx) == sh[:, 1:2])
    inc = np.diff(sh, axis=0, prepend=sh[:1])
    inc_dy = res[T * 9:, 0].reshape(T, 9)
    inc_dx = res[T * 9:, 1].reshape(T, 9)
    # (an individual 121-px ROI can lose a 35-px jump to noise -- the reference does too, see the oracle
    #  spot checks above -- so the truth check for "inc" is on the grid median, as speckle_stack_stats uses it)
    assert np.all(np.median(np.rint(inc_dy), axis=1) == inc[:, 0]) and np.all(np.median(np.rint(inc_dx), axis=1) == inc[:, 1])


@pytest.mark.parametrize("rm", [True, False])
@pytest.mark.parametrize("nm", ["peak", "none"])
def test_xcorr2d(gs, rm, nm):
    from oracle import signal_np as S

    a = synth.speckle_frame(512, 21)
    b = np.roll(a, (5, -9), axis=(0, 1)) + synth.speckle_frame(512, 22) * 0.1
    got, xl, yl = gs.xcorr2d(a, b, remove_mean=rm, normalize=nm, dx=0.5, dy=0.25)
    ref, xlr, ylr = S.xcorr2d(a.astype(np.float64), b.astype(np.float64), remove_mean=rm, normalize=nm, dx=0.5, dy=0.25)
    ref = np.real(ref)
    assert float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))) < 1e-5
    np.testing.assert_array_equal(xl, xlr)
    np.testing.assert_array_equal(yl, ylr)
    if rm:
        i, j = np.unravel_index(np.argmax(got), got.shape)
        assert (i - 256, j - 256) == (-5, 9)      # corr(a, roll(a)) peaks at minus the shift
    st = gs.xcorr2d(a, b, remove_mean=rm, standardize=True, normalize=nm)[0]
    str_ = np.real(S.xcorr2d(a.astype(np.float64), b.astype(np.float64), remove_mean=rm, standardize=True, normalize=nm)[0])
    assert float(np.max(np.abs(st - str_)) / np.max(np.abs(str_))) < 1e-5


@pytest.mark.parametrize("backend", ["opencv", "skimage"])
def test_template_matching_vs_oracle(gs, backend, observe):
    """NCC template matching (signal/tracking.py:81-188) against the float64 oracle (parity with cv2 / scikit-image is
    unpinned).  Integer arg-max exact; peak, sub-pixel shift and snr at float32 rounding level (observed <= 5e-7, bars 2e-6)."""
    from barc4dip_amd import synth
    from oracle import ncc_np as N

    n, h = 256, 41
    f0 = synth.speckle_frame(n, 21)
    for (dy, dx), (y0, x0) in (((3, -5), (100, 90)), ((-17, 22), (60, 140)), ((0, 0), (0, 0)), ((-9, -9), (n - h, n - h))):
        fr = np.roll(f0, (dy, dx), axis=(0, 1)) + np.random.default_rng(5).normal(size=(n, n)).astype(np.float32) * 20
        sl = (slice(y0, y0 + h), slice(x0, x0 + h))
        want = N.template_matching(f0[sl], fr, slices_yx=sl, backend=backend)
        got = gs.template_matching(f0[sl], fr, slices_yx=sl, backend=backend)
        wi = N.template_matching(f0[sl], fr, slices_yx=sl, backend=backend, subpixel=False)
        gi = gs.template_matching(f0[sl], fr, slices_yx=sl, backend=backend, subpixel=False)
        assert (gi[0], gi[1]) == (wi[0], wi[1])                       # integer part: bit-exact
        observe("ncc_256/peak_abs", abs(got[2] - want[2]), BARS["ncc_256/peak_abs"])
        observe("ncc_256/sub_px", max(abs(got[0] - want[0]), abs(got[1] - want[1])), BARS["ncc_256/sub_px"])
        observe("ncc_256/snr_rel", _rel(got[3], want[3]), BARS["ncc_256/snr_rel"])
    # template referenced to a position it was not cut from (slices None -> centred reference), odd size
    tpl = f0[30:71, 50:91]
    want = N.template_matching(tpl, f0, backend=backend)
    got = gs.template_matching(tpl, f0, backend=backend)
    observe("ncc_256/sub_px", max(abs(got[0] - want[0]), abs(got[1] - want[1])), BARS["ncc_256/sub_px"])
    with pytest.raises(ValueError):
        gs.template_matching(np.zeros((300, 10), np.float32), f0)
    with pytest.raises(ValueError):
        gs.template_matching(tpl, f0, backend="internal")


def test_template_matching_batch_map_and_truth(gs, observe):
    """Batched call: every (frame, ROI) pair recovers its integer shift; arg-max indices equal the oracle's."""
    from barc4dip_amd import synth
    from oracle import ncc_np as N

    stack, sh = synth.shifted_stack(4, 256, seed=77, max_shift=10)
    rois = [(40, 101, 50, 111), (120, 181, 130, 191), (10, 71, 180, 241)]
    pair_img = [t for t in range(4) for _ in rois]
    pair_tpl = [k for _ in range(4) for k in range(len(rois))]
    res, pij = gs.template_matching_batch(stack, stack, [0] * len(rois), rois, pair_img, pair_tpl, backend="skimage",
                                          subpixel=False, return_peak_ij=True)
    for i, (t, k) in enumerate(zip(pair_img, pair_tpl)):
        y0, y1, x0, x1 = rois[k]
        assert (res[i, 0], res[i, 1]) == (sh[t][0], sh[t][1])
        corr = N.match_template_ncc(stack[t], N.S.zscore2d(stack[0][y0:y1, x0:x1], 1e-9).astype(np.float32))
        assert tuple(pij[i]) == np.unravel_index(int(np.argmax(corr)), corr.shape)
        observe("ncc_batch/peak_abs", abs(res[i, 2] - float(corr.max())), BARS["ncc_batch/peak_abs"])


@pytest.mark.parametrize("shape", [(200, 300), (171, 170), (600, 720), (720, 600), (720, 1280), (1080, 1920)])   # the last four: mixed-radix kernels
def test_phase_correlation_general_sizes(gs, shape, observe):
    """Phase correlation on frames that are not a power of two (DFT-matrix / fused mixed-radix plans) against the
    oracle: integer arg-max exact, sub-pixel shift / peak within the float32 bar of the power-of-two path."""
    from oracle import signal_np as S

    H, W = shape
    rng = np.random.default_rng(H + W)
    base = synth_frame(max(H, W), 5)[:H, :W]
    for (dy, dx), (h, w) in (((3, -5), (61, 61)), ((-7, 11), (41, 81))):
        fr = (np.roll(base, (dy, dx), axis=(0, 1)) + rng.normal(size=(H, W)) * 20).astype(np.float32)
        y0, x0 = (H - h) // 2 - 10, (W - w) // 2 + 7
        sl = (slice(y0, y0 + h), slice(x0, x0 + w))
        want = S.phase_correlation(base[sl], fr, slices_yx=sl)
        got = gs.phase_correlation(base[sl], fr, slices_yx=sl)
        wi = S.phase_correlation(base[sl], fr, slices_yx=sl, subpixel=False)
        gi = gs.phase_correlation(base[sl], fr, slices_yx=sl, subpixel=False)
        assert (gi[0], gi[1]) == (wi[0], wi[1])            # integer output: equal to the reference's, whatever it finds
        if shape[0] <= 600:                                # (on the 720-row crops the reference itself loses some of these
            assert (gi[0], gi[1]) == (dy, dx)              #  61 / 41-px templates to noise: only parity is asserted there)
        observe("general_sizes/sub_px", max(abs(got[0] - want[0]), abs(got[1] - want[1])), BARS["general_sizes/sub_px"])
        observe("general_sizes/peak_rel", _rel(got[2], want[2]), BARS["general_sizes/peak_rel"])
        observe("general_sizes/snr_rel", _rel(got[3], want[3]), BARS["general_sizes/snr_rel"])


@pytest.mark.parametrize("shape", [(64, 64), (128, 256), (2048, 512), (512, 2048), (2048, 2048), (4096, 1024),
                                   (600, 720), (1080, 1920), (767, 1024), (1024, 768)])   # the last four: mixed-radix kernels (767: fused route)
def test_phase_correlation_power_of_two_sizes_all_routes(gs, shape, observe):
    """Every row-kernel geometry of the power-of-two route (1 to 32 row pairs per workgroup, 1 to 3 workgroups for the rows
    around the peak) and the mixed-radix route (quads of row pairs, odd heights) with the median expectation on (map-free pass
    + rows around the peak), off (full map) and deliberately wrong (gated full-map pass): bit-identical rows, and parity with
    the float64 oracle."""
    from barc4dip_amd import _ffi
    from oracle import signal_np as S

    H, W = shape
    rng = np.random.default_rng(H * 3 + W)
    base = synth_frame(max(H, W), 11)[:H, :W]
    big = 121 if max(H, W) >= 1024 else 41   # (a 41 x 61 template is lost in the noise of a 4096 x 1024 frame: the reference's own answer)
    h, w = min(big, H // 2 - 1) | 1, min(big + 20, W // 2 - 1) | 1
    dy, dx = max(-5, -(H // 8)), min(9, W // 8)
    fr = (np.roll(base, (dy, dx), axis=(0, 1)) + rng.normal(size=(H, W)) * 5).astype(np.float32)
    y0, x0 = (H - h) // 2 - 3, (W - w) // 2 + 2
    sl = (slice(y0, y0 + h), slice(x0, x0 + w))
    lib = _ffi.lib()
    got = {}
    try:
        for mode in (1, 0, 2):
            assert lib.b4d_set_option(b"track_predict_bin", mode) == 0
            got[mode] = (gs.phase_correlation(base[sl], fr, slices_yx=sl), gs.phase_correlation(base[sl], fr, slices_yx=sl, subpixel=False))
    finally:
        lib.b4d_set_option(b"track_predict_bin", 1)
    assert got[1] == got[0] == got[2]
    want = S.phase_correlation(base[sl], fr, slices_yx=sl)
    wi = S.phase_correlation(base[sl], fr, slices_yx=sl, subpixel=False)
    assert (got[1][1][0], got[1][1][1]) == (wi[0], wi[1]) == (dy, dx)
    observe("all_routes/sub_px", max(abs(got[1][0][0] - want[0]), abs(got[1][0][1] - want[1])), BARS["all_routes/sub_px"])
    observe("all_routes/peak_rel", _rel(got[1][0][2], want[2]), BARS["all_routes/peak_rel"])
    observe("all_routes/snr_rel", _rel(got[1][0][3], want[3]), BARS["all_routes/snr_rel"])


def synth_frame(n, seed):
    from barc4dip_amd import synth

    return synth.speckle_frame(n, seed)


def test_sharded_tracking_equals_single_rank(gs):
    """Two frame shards tracked separately (frame 0 + one-frame halo handed over explicitly, as the collective of
    metrics/sharded.py would deliver them) give exactly the numbers of one call over the whole stack."""
    from barc4dip_amd import synth
    from barc4dip_amd.metrics import sharded

    stack, sh = synth.shifted_stack(6, 256, seed=5, max_shift=8)
    rois = [(40, 101, 50, 111), (120, 181, 130, 191)]
    full = sharded.track_stack_sharded(stack, rois, frame0=stack[0], prev=stack[0])
    a = sharded.track_stack_sharded(stack[:2], rois, frame0=stack[0], prev=stack[0])
    b = sharded.track_stack_sharded(stack[2:], rois, frame0=stack[0], prev=stack[1])
    for k in full:
        assert np.array_equal(np.concatenate([a[k], b[k]]), full[k]), k
    # frames in blocks (bounded spectra workspace, metrics/sharded.py: track_abs_inc): a budget that fits ONE frame per
    # call -- (1 image + 2 templates + 2 abs templates) half spectra of 256 KiB -- and one that fits two
    for budget in (5 * 4 * 256 * 256, 8 * 4 * 256 * 256):
        blk = sharded.track_stack_sharded(stack, rois, frame0=stack[0], prev=stack[0], workspace_bytes=budget)
        for k in full:
            assert np.array_equal(blk[k], full[k]), (budget, k)
    tpl = sharded.track_stack_sharded(stack, rois, frame0=stack[0], prev=stack[0], method="template", backend="skimage")
    tpl1 = sharded.track_stack_sharded(stack, rois, frame0=stack[0], prev=stack[0], method="template", backend="skimage",
                                       workspace_bytes=9 * 4 * 256 * 256)
    for k in tpl:
        assert np.array_equal(tpl[k], tpl1[k]), k
    big = [(30, 151, 40, 161)]                                       # a well-conditioned ROI also recovers the ground truth
    tr = sharded.track_stack_sharded(stack, big, frame0=stack[0], prev=stack[0], subpixel=False)
    assert np.array_equal(tr["dy_abs"][:, 0], sh[:, 0]) and np.array_equal(tr["dx_abs"][:, 0], sh[:, 1])
    assert np.array_equal(tr["dy_inc"][1:, 0], np.diff(sh[:, 0])) and np.array_equal(tr["dx_inc"][1:, 0], np.diff(sh[:, 1]))


def test_template_matching_mixed_template_sizes(gs, observe):
    """One batched call with templates of different shapes (every pair has its own match-map geometry)."""
    from barc4dip_amd import synth
    from oracle import ncc_np as N

    stack, sh = synth.shifted_stack(3, 256, seed=21, max_shift=9)
    rois = [(40, 101, 50, 111), (100, 141, 30, 151), (5, 200, 120, 161)]     # 61x61, 41x121, 195x41
    pair_img = [t for t in range(3) for _ in rois]
    pair_tpl = [k for _ in range(3) for k in range(len(rois))]
    res, pij = gs.template_matching_batch(stack, stack, [0] * len(rois), rois, pair_img, pair_tpl, backend="opencv",
                                          subpixel=True, return_peak_ij=True)
    for i, (t, k) in enumerate(zip(pair_img, pair_tpl)):
        y0, y1, x0, x1 = rois[k]
        want = N.template_matching(stack[0][y0:y1, x0:x1], stack[t], slices_yx=(slice(y0, y1), slice(x0, x1)), backend="opencv")
        observe("ncc_mixed/sub_px", max(abs(res[i, 0] - want[0]), abs(res[i, 1] - want[1])), BARS["ncc_mixed/sub_px"])
        observe("ncc_mixed/peak_abs", abs(res[i, 2] - want[2]), BARS["ncc_mixed/peak_abs"])
        observe("ncc_mixed/snr_rel", _rel(res[i, 3], want[3]), BARS["ncc_mixed/snr_rel"])
        assert (round(res[i, 0]), round(res[i, 1])) == (sh[t][0], sh[t][1])


# Observed on MI355X (round 2, printed by the test below), 72 pairs of the 1024^2 protocol:
#   vs the float64 oracle        sub-pixel 4.5e-6 px, peak 6.5e-5 rel (1.8e-5 abs), snr 2.0e-4 rel
#   vs the float32 protocol      sub-pixel 4.2e-6 px, peak 1.2e-4 rel,             snr 2.5e-4 rel
# (the reference's own float32 and float64 paths differ by 1.2e-4 / 2.5e-4 in peak / snr: the whitened spectrum gives
# every bin unit modulus, so rounding noise in weak bins is not damped).  Tolerances = about 2 x the observed maxima;
# SURVEY.md §8(d)'s 1e-4 absolute holds for the sub-pixel shift and the peak, snr (values 50 ... 800) is held relative.
CFG3_TOL_SUB_PX = 1e-5      # |dy|, |dx| vs the float64 oracle
CFG3_TOL_PEAK_REL = 1.5e-4
CFG3_TOL_SNR_REL = 4e-4


def test_cfg3_size(gs):
    """BASELINE.json config 3 at its stated frame size: 1024 x 1024 stack, SURVEY.md §8(d) protocol (3 x 3 ROI grid,
    "abs" + "inc" templates of 121 px), T = 4.  EVERY pair: arg-max index and integer shift equal to the oracle's
    (signal/tracking.py:283-290), sub-pixel / peak / snr against the float64 oracle AND against the reference's own
    float32 protocol (the oracle run on the float32 frames, which is what the device mirrors)."""
    from barc4dip_amd.geometry import roi_grid_3x3
    from oracle import signal_np as S

    T, n, side = 4, 1024, 121
    stack, sh = synth.shifted_stack(T, n, seed=1234, max_shift=32)
    grid, _ = roi_grid_3x3((n, n), (side, side), (side // 2, side // 2))
    rois = [(s[0].start, s[0].stop, s[1].start, s[1].stop) for s in grid.ravel()]
    tpl_frame = [0] * 9 + [max(t - 1, 0) for t in range(T) for _ in range(9)]
    tpl_roi = rois + rois * T
    pair_img = [t for t in range(T) for _ in range(9)] * 2
    pair_tpl = [k for _ in range(T) for k in range(9)] + [9 + 9 * t + k for t in range(T) for k in range(9)]
    res, pij = gs.phase_correlation_batch(stack, stack, tpl_frame, tpl_roi, pair_img, pair_tpl, return_peak_ij=True)
    npairs = len(pair_img)
    assert npairs == 2 * 9 * T and res.shape == (npairs, 4)
    worst = {"sub64": 0.0, "peak64": 0.0, "snr64": 0.0, "sub32": 0.0, "peak32": 0.0, "snr32": 0.0, "peak64abs": 0.0}
    for i in range(npairs):
        r = rois[i % 9]
        sl = (slice(r[0], r[1]), slice(r[2], r[3]))
        tpl, img = stack[tpl_frame[pair_tpl[i]]][sl], stack[pair_img[i]]
        ref64 = S.phase_correlation(tpl.astype(np.float64), img.astype(np.float64), slices_yx=sl)
        ref32 = S.phase_correlation(tpl, img, slices_yx=sl)              # float32 in -> complex64 transforms, like the reference
        mag = S.phase_correlation_map(tpl.astype(np.float64), img.astype(np.float64), slices_yx=sl)
        mi, mj = np.unravel_index(np.argmax(mag), mag.shape)
        assert (int(pij[i, 0]), int(pij[i, 1])) == (int(mi), int(mj)), i           # index output: bit exact
        assert round(res[i, 0]) == round(ref64[0]) and round(res[i, 1]) == round(ref64[1]), i
        worst["peak64abs"] = max(worst["peak64abs"], abs(res[i, 2] - ref64[2]))
        for tag, ref in (("64", ref64), ("32", ref32)):
            worst["sub" + tag] = max(worst["sub" + tag], abs(res[i, 0] - ref[0]), abs(res[i, 1] - ref[1]))
            worst["peak" + tag] = max(worst["peak" + tag], abs(res[i, 2] - ref[2]) / abs(ref[2]))
            worst["snr" + tag] = max(worst["snr" + tag], abs(res[i, 3] - ref[3]) / abs(ref[3]))
    print("cfg3 1024^2 observed maxima:", {k: f"{v:.2e}" for k, v in worst.items()})
    assert worst["sub64"] <= CFG3_TOL_SUB_PX and worst["peak64"] <= CFG3_TOL_PEAK_REL and worst["snr64"] <= CFG3_TOL_SNR_REL
    assert worst["peak64abs"] <= 1e-4                     # peak within 1e-4 absolute
    # against the reference's float32 protocol (what the device mirrors): the same order as float32-vs-float64 of the reference
    assert worst["sub32"] <= 1e-5 and worst["peak32"] <= 2.5e-4 and worst["snr32"] <= 5e-4
    # ground truth of the protocol: abs shifts = the imposed spiral on every ROI
    assert np.all(np.rint(res[:9 * T, 0]).reshape(T, 9) == sh[:, 0:1]) and np.all(np.rint(res[:9 * T, 1]).reshape(T, 9) == sh[:, 1:2])


def test_median_bin_prediction_is_only_a_route(gs):
    """b4d_phase_correlation gathers the EXPECTED median bin of every |corr| map in the pass that produces it (whitened maps:
    median ~ 0.6745 / sqrt(N)) and, with an expectation, does not store the map at all: three row pairs around the peak are
    recomputed for the Taylor step, and only pairs whose histogram disagrees get their full map from a gated second pass.
    All routes must give bit-identical rows -- expectation off (every pair on the full map written by the first pass),
    expectation on, expectation deliberately wrong (every pair through the gated pass) -- and an un-whitened-looking input
    (a constant template region: degenerate map) must survive a wrong expectation."""
    from barc4dip_amd import _ffi

    stack, sh = synth.shifted_stack(4, 512, seed=77, max_shift=12)
    rois = [(100, 221, 90, 211), (300, 421, 280, 401)]
    tpl_frame = [0, 0] + [max(t - 1, 0) for t in range(4) for _ in range(2)]
    tpl_roi = rois + rois * 4
    pair_img = [t for t in range(4) for _ in range(2)] * 2
    pair_tpl = [k for _ in range(4) for k in range(2)] + [2 + 2 * t + k for t in range(4) for k in range(2)]
    flat = stack.copy()
    flat[1, 100:221, 90:211] = 7.0            # a constant ROI: z-score 0 -> all-zero cross spectrum -> the map is not Rayleigh
    lib = _ffi.lib()
    out = {}
    try:
        for mode in (1, 0, 2):   # 2: a deliberately wrong bin -- the map-free pass, then the gated full-map pass for every pair
            assert lib.b4d_set_option(b"track_predict_bin", mode) == 0
            out[mode] = (gs.phase_correlation_batch(stack, stack, tpl_frame, tpl_roi, pair_img, pair_tpl, return_peak_ij=True),
                         gs.phase_correlation_batch(flat, flat, tpl_frame, tpl_roi, pair_img, pair_tpl, return_peak_ij=True))
    finally:
        lib.b4d_set_option(b"track_predict_bin", 1)
    for m in (1, 2):
        for a, b in zip(out[m], out[0]):
            assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1])
    assert lib.b4d_set_option(b"no_such_option", 1) != 0
    assert np.all(np.rint(out[1][0][0][:8, 0]).reshape(4, 2) == sh[:, 0:1])


@pytest.mark.parametrize("shape", [(256, 256), (200, 300), (512, 1024)])
def test_phase_correlation_skimage_backend_vs_oracle(gs, shape):
    """``backend="skimage"`` (tracking.py:262-272): device transforms + host refinement on the 0.1-px grid against the float64
    restatement of skimage.registration.phase_cross_correlation (oracle/phase_skimage_np.py; parity with scikit-image itself is
    unpinned).  The shift is a multiple of 0.1 px: equal to the oracle's, peak and snr NaN like the reference."""
    from oracle import phase_skimage_np as P

    H, W = shape
    base = synth_frame(max(H, W), 9)[:H, :W].astype(np.float64)
    ky, kx = np.fft.fftfreq(H)[:, None], np.fft.fftfreq(W)[None, :]
    for (dy, dx), (h, w) in (((3.3, -5.7), (121, 121)), ((-7.5, 11.25), (61, 81)), ((0.0, 0.0), (91, 91))):
        fr = np.real(np.fft.ifft2(np.fft.fft2(base) * np.exp(-2j * np.pi * (ky * dy + kx * dx)))).astype(np.float32)
        y0, x0 = (H - h) // 2 - 6, (W - w) // 2 + 9
        sl = (slice(y0, y0 + h), slice(x0, x0 + w))
        tpl = base[sl].astype(np.float32)
        for sub in (True, False):
            got = gs.phase_correlation(tpl, fr, slices_yx=sl, backend="skimage", subpixel=sub)
            want = P.phase_correlation_skimage(tpl, fr, slices_yx=sl, subpixel=sub)
            assert np.isnan(got[2]) and np.isnan(got[3])
            assert abs(got[0] - want[0]) < 1e-9 and abs(got[1] - want[1]) < 1e-9, (shape, (dy, dx), sub, got, want)
            assert abs(got[0] - dy) <= (0.15 if sub else 0.5) and abs(got[1] - dx) <= (0.15 if sub else 0.5)
    # the dispatcher reaches the same route; float64 frames keep float64 grid values (the reference's complex128 branch)
    via = gs.track_translation(tpl, fr, slices_yx=sl, method="phase", backend="skimage", subpixel=False)
    assert via[:2] == got[:2]
    g64 = gs.phase_correlation(tpl.astype(np.float64), fr.astype(np.float64), slices_yx=sl, backend="skimage")
    w64 = P.phase_correlation_skimage(tpl.astype(np.float64), fr.astype(np.float64), slices_yx=sl)
    assert g64[:2] == w64[:2]


@pytest.mark.parametrize("backend", ["opencv", "skimage"])
def test_template_matching_median_guess_is_only_a_route(gs, backend):
    """b4d_template_match guesses the bin of the median of |NCC map| from 4096 samples, counts and gathers that bin while the map
    is written and finishes the select on the gathered values; pairs whose middle rank falls elsewhere (and every pair with
    "track_predict_bin" 0) take the whole select on their map.  Both routes: bit-identical rows and arg-max indices -- on speckle
    frames, on a frame with a constant region (maps full of exact zeros: the guess fails there) and on a non-power-of-two frame."""
    from barc4dip_amd import _ffi

    stack, sh = synth.shifted_stack(4, 512, seed=21, max_shift=10)
    flat = stack.copy()
    flat[:, 200:, :] = 5.0                                  # constant lower part: zero-variance windows -> r = 0 exactly
    odd = stack[:, :300, :417].copy()
    rois = [(100, 161, 90, 171), (30, 121, 280, 341)]
    tpl_frame = [0, 0] + [max(t - 1, 0) for t in range(4) for _ in range(2)]
    tpl_roi = rois + rois * 4
    pair_img = [t for t in range(4) for _ in range(2)] * 2
    pair_tpl = [k for _ in range(4) for k in range(2)] + [2 + 2 * t + k for t in range(4) for k in range(2)]
    lib = _ffi.lib()
    out = {}
    try:
        for mode in (1, 0):
            assert lib.b4d_set_option(b"track_predict_bin", mode) == 0
            out[mode] = [gs.template_matching_batch(x, x, tpl_frame, tpl_roi, pair_img, pair_tpl, backend=backend, return_peak_ij=True)
                         for x in (stack, flat, odd)]
    finally:
        lib.b4d_set_option(b"track_predict_bin", 1)
    for a, b in zip(out[1], out[0]):
        assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1])
    assert np.all(np.rint(out[1][0][0][:8, 0]).reshape(4, 2) == sh[:, 0:1])
