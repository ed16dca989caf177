"""dev: second fuzz round -- tracking, stacks, preprocessing on awkward inputs vs the oracle."""
import sys
import traceback
import warnings

import numpy as np

sys.path.insert(0, ".")
from barc4dip_amd import metrics as gm, signal as gs, preprocessing as gp, synth  # noqa: E402
from oracle import metrics_np as M, signal_np as S, wiener_np as W, preprocess_np as P, ncc_np as N  # noqa: E402

warnings.simplefilter("ignore")
rng = np.random.default_rng(1)


def both(f, g, tag):
    r = []
    for fn in (f, g):
        try:
            r.append((fn(), None))
        except Exception as e:  # noqa: BLE001
            r.append((None, type(e).__name__ + ": " + str(e)[:60]))
    if (r[0][1] is None) != (r[1][1] is None) or (r[0][1] and r[0][1].split(":")[0] != r[1][1].split(":")[0]):
        print("MISMATCH exception", tag, "| gpu:", r[0][1], "| oracle:", r[1][1])
    return r[0][0], r[1][0]


f0 = synth.speckle_frame(256, 9)
f1 = np.roll(f0, (4, -6), axis=(0, 1))
# --- tracking
for tag, tpl, img, kw in (
    ("border_roi", f0[0:61, 195:256], f1, dict(slices_yx=(slice(0, 61), slice(195, 256)))),
    ("tpl_eq_img", f0, f1, dict(slices_yx=(slice(0, 256), slice(0, 256)))),
    ("tpl_too_big", np.zeros((300, 300), np.float32), f1, {}),
    ("even_tpl_noslices", f0[:60, :60], f1, {}),
    ("u16_inputs", f0[100:161, 100:161].astype(np.uint16), f1.astype(np.uint16), dict(slices_yx=(slice(100, 161), slice(100, 161)))),
    ("f64_inputs", f0[100:161, 100:161].astype(np.float64), f1.astype(np.float64), dict(slices_yx=(slice(100, 161), slice(100, 161)))),
    ("const_tpl", np.full((41, 41), 3.0, np.float32), f1, dict(slices_yx=(slice(10, 51), slice(10, 51)))),
):
    for name, gf, of in (("phase", gs.phase_correlation, S.phase_correlation), ("template", gs.template_matching, N.template_matching)):
        a, b = both(lambda: gf(tpl, img, **kw), lambda: of(tpl, img, **kw), f"{tag}/{name}")
        if a is not None and b is not None:
            ok = np.allclose(a[:2], b[:2], atol=2e-2, equal_nan=True) and np.isclose(a[2], b[2], rtol=2e-3, atol=2e-4, equal_nan=True)
            if not ok:
                print("MISMATCH", tag, name, a, b)
# --- stacks
st = np.stack([synth.speckle_frame(256, 20 + i) for i in range(3)])
for tag, fn_g, fn_o, arg, kw in (
    ("sharp_stack_T1", gm.sharpness_stack_stats, M.sharpness_stack_stats, st[:1], dict(metrics=("gradient", "stats"))),
    ("speckle_stack_u16", gm.speckle_stack_stats, M.speckle_stack_stats, st.astype(np.uint16),
     dict(metrics=("stats",), roi_grain_factor=10.0, tracking_method="phase", tracking_backend="internal")),
    ("speckle_stack_2d", gm.speckle_stack_stats, M.speckle_stack_stats, st[0], {}),
):
    a, b = both(lambda: fn_g(arg, verbose=False, **kw), lambda: fn_o(arg, **kw), tag)
    if a is not None and b is not None:
        ta, tb = a.get("temporal", {}), b.get("temporal", {})
        for blk in tb:
            if isinstance(tb[blk], dict):
                for k, v in tb[blk].items():
                    if isinstance(v, np.ndarray) and not np.allclose(ta[blk][k], v, atol=2e-2):
                        print("MISMATCH", tag, blk, k, ta[blk][k], v)
# --- preprocessing
img = synth.speckle_frame(200, 5)[:150, :180]
for tag, kw in (("sigma_tuple", dict(sigma=(0.8, 1.7))), ("sigma_bad", dict(sigma=-1.0)), ("sigma_three", dict(sigma=(1, 2, 3))),
                ("balance", dict(sigma=1.0, balance=0.2)), ("noclip", dict(sigma=1.0, clip=False)), ("rl", dict(sigma=1.0, method="rl", num_iter=5)),
                ("bad_method", dict(sigma=1.0, method="foo")), ("bad_pad", dict(sigma=1.0, pad_mode="edge"))):
    a, b = both(lambda: gp.deconvolve_psf(img, **kw), lambda: W.deconvolve_psf(img, **{k: v for k, v in kw.items()}), "deconv/" + tag)
    if a is not None and b is not None and not (np.max(np.abs(a - b)) <= 2e-5 * np.max(np.abs(img))):
        print("MISMATCH deconv", tag, float(np.max(np.abs(a - b))))
nanimg = img.copy(); nanimg[5, 7] = np.nan
a, b = both(lambda: gp.deconvolve_psf(nanimg, sigma=1.0), lambda: W.deconvolve_psf(nanimg, sigma=1.0), "deconv/nan")
if a is not None and b is not None and not np.array_equal(np.isnan(a), np.isnan(b)):
    print("MISMATCH deconv nan pattern", int(np.isnan(a).sum()), int(np.isnan(b).sum()))
zimg = np.zeros_like(img)
a, b = both(lambda: gp.deconvolve_psf(zimg, sigma=1.0), lambda: W.deconvolve_psf(zimg, sigma=1.0), "deconv/zeros")
if a is not None and not np.array_equal(a, b):
    print("MISMATCH deconv zeros")
flats = rng.poisson(2000, size=(3, 150, 180)).astype(np.float32); darks = rng.poisson(100, size=(150, 180)).astype(np.float32)
flats[:, 3, 4] = np.nan
for tag, kw in (("nanflat", dict(flats=flats, darks=darks)), ("nanflat_repair", dict(flats=flats, darks=darks, bad_pixel_removal=True)),
                ("shape_mismatch", dict(flats=flats[:, :100], darks=darks)), ("4d", dict(flats=flats[None]))):
    a, b = both(lambda: gp.flat_field_correction(img, **kw), lambda: P.flat_field_correction(img, **kw), "flat/" + tag)
    if a is not None and b is not None and not np.array_equal(a, b, equal_nan=True):
        print("MISMATCH flat", tag, int((a != b).sum()))
print("fuzz2 done")
