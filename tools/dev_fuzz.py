"""dev: API-level comparison of the GPU path with the oracle on awkward inputs (dtypes, strides, sizes, NaNs)."""
import sys
import traceback
import warnings

import numpy as np

sys.path.insert(0, ".")
from barc4dip_amd import metrics as gm, signal as gs, preprocessing as gp, synth  # noqa: E402
from oracle import metrics_np as M, signal_np as S  # noqa: E402

warnings.simplefilter("ignore")
rng = np.random.default_rng(0)
base = synth.speckle_frame(600, 3)


def same_exc(f, g, tag):
    try:
        a = f()
        ea = None
    except Exception as e:  # noqa: BLE001
        a, ea = None, type(e).__name__
    try:
        b = g()
        eb = None
    except Exception as e:  # noqa: BLE001
        b, eb = None, type(e).__name__
    if ea != eb:
        print("MISMATCH exception", tag, "gpu:", ea, "oracle:", eb)
    return a, b


def cmp_dict(a, b, tag, rtol=2e-5):
    if a is None or b is None:
        return
    for k, v in b.items():
        if isinstance(v, dict):
            cmp_dict(a[k], v, tag + "/" + k, rtol)
        elif isinstance(v, (float, int, np.floating)) and k in a:
            x = float(a[k])
            if not (np.isclose(x, float(v), rtol=rtol, atol=1e-9, equal_nan=True)):
                print("MISMATCH", tag, k, x, float(v))
        elif isinstance(v, np.ndarray) and v.dtype.kind == "f" and k in a and np.asarray(a[k]).shape == v.shape and k not in ("autocorr",):
            if not np.allclose(np.asarray(a[k], float), v, rtol=rtol, atol=1e-9, equal_nan=True):
                print("MISMATCH array", tag, k, float(np.nanmax(np.abs(np.asarray(a[k], float) - v))))


cases = {
    "u16_384": (base[:384, :384]).astype(np.uint16),
    "f64_300x420": base[:300, :420].astype(np.float64),
    "noncontig": base[::2, ::2][:290, :290],
    "fortran": np.asfortranarray(base[:256, :256]),
    "int32_130": base[:130, :130].astype(np.int32),
    "f32_600": base,
    "withnan_512": np.where(rng.random((512, 512)) < 1e-4, np.nan, base[:512, :512]).astype(np.float32),
    "const_256": np.full((256, 256), 7.0, np.float32),
    "zeros_256": np.zeros((256, 256), np.float32),
}
for name, img in cases.items():
    try:
        a, b = same_exc(lambda: gm.speckle_stats(img, verbose=False), lambda: M.speckle_stats(img), name + "/speckle_stats")
        if a and b:
            a["full"].get("grain", {}).pop("autocorr", None); b["full"].get("grain", {}).pop("autocorr", None)
            cmp_dict(a, b, name + "/speckle")
        a, b = same_exc(lambda: gm.sharpness_stats(img, verbose=False), lambda: M.sharpness_stats(img), name + "/sharpness_stats")
        cmp_dict(a, b, name + "/sharpness")
        for fn in ("fft2d", "psd2d", "autocorr2d"):
            a, b = same_exc(lambda: getattr(gs, fn)(img), lambda: getattr(S, fn)(np.asarray(img, dtype=np.float64) if img.dtype.kind != "f" else img),
                            name + "/" + fn)
            if a is not None and b is not None:
                x, y = np.asarray(a[0]), np.asarray(b[0])
                den = np.nanmax(np.abs(y)) or 1.0
                err = np.nanmax(np.abs(x - y)) / den if np.isfinite(y).all() else 0.0
                if not (err < 2e-5) or x.dtype != y.dtype:
                    print("MISMATCH", name, fn, err, x.dtype, y.dtype)
    except Exception:  # noqa: BLE001
        print("CRASH", name)
        traceback.print_exc()
print("fuzz done")
