"""Generate tests/golden/*.npz by RUNNING THE REAL REFERENCE (build container only).

    python oracle/make_golden.py

Every array written here is data: seeded synthetic inputs (or their generator
parameters) and the outputs the reference produced for them.  No reference source
travels.  NumPy 2.2.6 / SciPy 1.15.3 were used; versions are recorded in each file.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from barc4dip_amd import synth  # noqa: E402
from oracle import load_reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def flat(prefix, d, out):
    """Flatten nested dicts of scalars/arrays into 'a/b/c' keys."""
    for k, v in d.items():
        key = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            flat(key, v, out)
        elif isinstance(v, (tuple, list)) and all(isinstance(t, (int, float, np.integer, np.floating)) for t in v):
            out[key] = np.asarray(v)
        elif isinstance(v, (str, bool)) or v is None:
            continue
        elif isinstance(v, np.ndarray) and v.dtype == object:
            continue
        else:
            out[key] = np.asarray(v)
    return out


def survey_kat_image(n=512):
    """The KAT input recorded in SURVEY.md §8c (noise-free speckle, pupil radius n/16)."""
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[-n // 2:n // 2, -n // 2:n // 2]
    pupil = (xx ** 2 + yy ** 2) <= (n / 16) ** 2
    field = np.fft.ifft2(np.fft.ifftshift(pupil * np.exp(2j * np.pi * rng.random((n, n)))))
    img = np.abs(field) ** 2
    return (img / img.mean() * 1000).astype(np.float32)


def main():
    ref = load_reference.load()
    os.makedirs(OUT, exist_ok=True)
    ver = np.array([np.__version__, scipy.__version__])

    # ------------------------------------------------------------------ signal: small dense cases
    rng = np.random.default_rng(42)
    g = {"versions": ver}
    cases = {
        "f64_24x32": rng.normal(size=(24, 32)) + 3.0,
        "f32_32x16": (rng.normal(size=(32, 16)) * 5 + 20).astype(np.float32),
        "f64_17x23": rng.random((17, 23)),
        "f32_64": synth.speckle_frame(64, 1234),
    }
    for name, a in cases.items():
        b = np.roll(a, (3, -5), axis=(0, 1)) + rng.normal(size=a.shape).astype(a.dtype) * 0.1
        g[f"{name}/a"] = a
        g[f"{name}/b"] = b
        F, fx, fy = ref.signal.fft2d(a, dx=0.5, dy=2.0)
        g[f"{name}/fft2d"] = F
        g[f"{name}/fx"] = fx
        g[f"{name}/fy"] = fy
        g[f"{name}/ifft2d"] = ref.fft.ifft2d(F)
        g[f"{name}/psd2d"] = ref.signal.psd2d(a)[0]
        g[f"{name}/psd2d_cal"] = ref.signal.psd2d(a, dx=0.5, dy=2.0)[0]
        g[f"{name}/psd2d_noscale"] = ref.signal.psd2d(a, scale=False)[0]
        big = a.size > 2048
        for rm in (True, False):
            for st in (True, False):
                for nm in ("peak", "none"):
                    if big and (rm, st, nm) not in ((True, False, "peak"), (True, True, "none")):
                        continue
                    tag = f"rm{int(rm)}_st{int(st)}_{nm}"
                    c, xl, yl = ref.signal.xcorr2d(a, b, remove_mean=rm, standardize=st, normalize=nm)
                    g[f"{name}/xcorr2d_{tag}"] = c
                    ac, _, _ = ref.signal.autocorr2d(a, remove_mean=rm, standardize=st, normalize=nm)
                    g[f"{name}/autocorr2d_{tag}"] = ac
        g[f"{name}/xlag"] = xl
        g[f"{name}/ylag"] = yl
    # 1-D helpers
    s1 = rng.normal(size=100) + 1.0
    s2 = np.roll(s1, 7) + 0.05 * rng.normal(size=100)
    g["1d/a"], g["1d/b"] = s1, s2
    g["1d/fft1d"], g["1d/fx"] = ref.signal.fft1d(s1, dx=0.25)
    g["1d/ifft1d"] = ref.fft.ifft1d(g["1d/fft1d"])
    g["1d/psd1d"] = ref.signal.psd1d(s1, dx=0.25)[0]
    g["1d/psd1d_noscale"] = ref.signal.psd1d(s1, scale=False)[0]
    g["1d/xcorr1d"], g["1d/lag"] = ref.corr.xcorr1d(s1, s2)
    g["1d/autocorr1d"] = ref.corr.autocorr1d(s1, standardize=True)[0]
    xs = np.linspace(0.0, 9.9, 100)
    g["1d/freq_axis_x"] = ref.signal.freq_axis1d(n=100, x=xs)
    g["1d/xs"] = xs
    np.savez_compressed(os.path.join(OUT, "signal_small.npz"), **g)

    # ------------------------------------------------------------------ tracking
    t = {"versions": ver}
    n = 256
    i0 = synth.speckle_intensity(n, 1234)
    frame0 = np.random.default_rng(1).poisson(i0).astype(np.float32)
    shifts = [(0, 0), (3, -5), (-17, 9), (32, 32), (-32, 31), (1, 0)]
    rows = []
    for k, (sy, sx) in enumerate(shifts):
        fr = np.random.default_rng(100 + k).poisson(np.roll(i0, (sy, sx), axis=(0, 1))).astype(np.float32)
        for side, cyx in ((121, None), (63, (100, 140)), (255, None)):
            sl = ref.roi.roi_slices((n, n), (side, side), center_yx=cyx)
            for sub in (True, False):
                for dt in (np.float32, np.float64):
                    r = ref.signal.phase_correlation(frame0[sl].astype(dt), fr.astype(dt), slices_yx=sl,
                                                     subpixel=sub)
                    rows.append([k, sy, sx, side, -1 if cyx is None else cyx[0], -1 if cyx is None else cyx[1],
                                 int(sub), 32 if dt is np.float32 else 64, *r])
    t["phase/rows"] = np.asarray(rows, dtype=np.float64)
    t["phase/cols"] = np.array(["frame", "sy", "sx", "side", "cy", "cx", "subpixel", "bits", "dy", "dx", "peak", "snr"])
    t["phase/seed_note"] = np.array(["i0=speckle_intensity(256,1234); frame0=default_rng(1).poisson(i0);"
                                    " frame k=default_rng(100+k).poisson(roll(i0,(sy,sx)))"])
    # centred default ROI via track_translation dispatcher
    fr = np.random.default_rng(101).poisson(np.roll(i0, (3, -5), axis=(0, 1))).astype(np.float32)
    sl = ref.roi.roi_slices((n, n), (121, 121))
    t["track/default"] = np.asarray(ref.signal.track_translation(frame0[sl], fr))
    # Taylor refinement on random 3x3 neighbourhoods
    tr = np.random.default_rng(7)
    nb = tr.random((32, 5, 5))
    nb[:, 2, 2] += 1.0
    t["taylor/in"] = nb
    t["taylor/out"] = np.asarray([ref.tracking._peak_subpixel_taylor(m, peak_ij=(2, 2)) for m in nb])
    # correlation magnitude map on a small case (full array)
    small0 = synth.speckle_frame(64, 11)
    small1 = np.roll(small0, (2, -3), axis=(0, 1))
    sls = ref.roi.roi_slices((64, 64), (31, 31))
    t["map64/f0"], t["map64/f1"] = small0, small1
    t["map64/result"] = np.asarray(ref.signal.phase_correlation(small0[sls], small1, slices_yx=sls))
    np.savez_compressed(os.path.join(OUT, "tracking.npz"), **t)

    # ------------------------------------------------------------------ metrics on 512^2
    m = {"versions": ver}
    kat = survey_kat_image(512)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, img in (("kat512", kat), ("poisson512", synth.speckle_frame(512, 1234))):
            for origin in ("lower", "upper"):
                flat(f"{tag}/{origin}/speckle", ref.metrics.speckle_stats(img, display_origin=origin, verbose=False), m)
                flat(f"{tag}/{origin}/sharpness", ref.metrics.sharpness_stats(img, display_origin=origin, verbose=False), m)
        # drop the big autocorr maps; keep a strided sample + centre cut
        for k in [k for k in m if k.endswith("grain/autocorr")]:
            ac = m.pop(k)
            m[k + "_cut_x"] = ac[ac.shape[0] // 2, :]
            m[k + "_cut_y"] = ac[:, ac.shape[1] // 2]
        # individual metric functions on awkward shapes / dtypes
        odd = synth.speckle_frame(256, 77)[:171, :200].astype(np.float64)
        m["odd/in_seed"] = np.array([256, 77, 171, 200])
        flat("odd/amplitude", ref.speckles.amplitude(odd), m)
        gr = ref.speckles.grain(odd)
        flat("odd/grain", {k: gr[k] for k in ("lx", "ly", "leq", "r")}, m)
        flat("odd/grain_binned", {k: ref.speckles.grain(odd, radial_method="binned")[k] for k in ("lx", "ly", "leq", "r")}, m)
        flat("odd/bandwidth", ref.speckles.bandwidth(odd), m)
        flat("odd/tenengrad", ref.sharpness.tenengrad(odd), m)
        m["odd/laplacian_variance"] = np.float64(ref.sharpness.laplacian_variance(odd))
        m["odd/spectral_entropy"] = np.float64(ref.sharpness.spectral_entropy(odd))
        flat("odd/inverse_autocorr_width", ref.sharpness.inverse_autocorr_width(odd), m)
        flat("odd/eigenvalues", ref.sharpness.eigenvalues(odd), m)
        flat("odd/moments", ref.metrics.distribution_moments(odd, saturation_value=3000.0), m)
        withnan = odd.copy()
        withnan[5, 7] = np.nan
        withnan[100, 3] = np.inf
        flat("nan/moments", ref.metrics.distribution_moments(withnan), m)
        flat("nan/tenengrad", ref.sharpness.tenengrad(np.where(np.isfinite(withnan), withnan, 0.0)), m)
        # maths helpers
        prof = np.exp(-0.5 * ((np.arange(101) - 50.3) / 6.0) ** 2)
        m["maths/profile"] = prof
        m["maths/width"] = np.asarray(ref.mstats.width_at_fraction(prof)[0])
        m["maths/width_half"] = np.asarray(ref.mstats.width_at_fraction(prof, fraction=0.5, center_index=50)[0])
        m["maths/dist"] = np.asarray(ref.mstats.distance_at_fraction_from_peak(prof[50:], fraction=0.5)[0])
        m["maths/width_edge"] = np.asarray(ref.mstats.width_at_fraction(np.ones(10) + np.arange(10))[0])
        acs = ref.signal.autocorr2d(synth.speckle_frame(128, 5))[0]
        m["maths/radial_interp"] = ref.radial.radial_mean_interpolated(acs)[0]
        m["maths/radial_binned"] = ref.radial.radial_mean_binned(acs)[0]
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), **m)

    # ------------------------------------------------------------------ stack (phase/internal tracker)
    s = {"versions": ver}
    stack, sh = synth.shifted_stack(5, 384, seed=1234, max_shift=12)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        res = ref.metrics.speckle_stack_stats(stack, metrics=("amplitude", "grain", "stats"), tiles=True,
                                              roi_grain_factor=24.0, tracking_method="phase",
                                              tracking_backend="internal", verbose=False, parallel=False)
        res["full"]["grain"].pop("autocorr")
        flat("speckle", {k: res[k] for k in ("full", "tiles", "temporal")}, s)
        s["speckle/meta/roi_size_yx"] = np.asarray(res["meta"]["tracking"]["roi_size_yx"])
        s["speckle/meta/roi_step_yx"] = np.asarray(res["meta"]["tracking"]["roi_step_yx"])
        flat("speckle/meta/grain0", res["meta"]["grain0"], s)
        res2 = ref.metrics.sharpness_stack_stats(stack[:3], metrics=("gradient", "laplacian", "spectral"),
                                                 verbose=False, parallel=False)
        flat("sharpness", {k: res2[k] for k in ("full", "tiles")}, s)
    s["shifts"] = sh
    np.savez_compressed(os.path.join(OUT, "stack.npz"), **s)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
