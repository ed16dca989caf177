"""dev: randomised soak of speckle_stats / sharpness_stats (random frame shapes, tiles on/off, origins) vs the oracle."""
import sys, time, warnings
sys.path.insert(0, ".")
import numpy as np
from barc4dip_amd import metrics as gm, synth
from oracle import metrics_np as M

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
RTOL = float(sys.argv[3]) if len(sys.argv) > 3 else 5e-5
bad = 0
worst = (0.0, "")


def walk(a, b, path, out):
    for k, v in b.items():
        if isinstance(v, dict):
            walk(a[k], v, path + "/" + k, out)
        elif isinstance(v, (float, int, np.floating)) or (isinstance(v, np.ndarray) and v.dtype.kind == "f" and k != "autocorr"):
            x, y = np.asarray(a[k], float), np.asarray(v, float)
            if x.shape != y.shape:
                out.append((path + "/" + k, "shape", x.shape, y.shape)); continue
            nanmis = np.isnan(x) != np.isnan(y)
            with np.errstate(invalid="ignore", divide="ignore"):
                rel = np.abs(x - y) / np.maximum(np.abs(y), 1e-9)
            rel = np.where(np.isnan(y), 0.0, rel)
            m = float(np.nanmax(rel)) if rel.size else 0.0
            if nanmis.any() or not m <= RTOL:
                out.append((path + "/" + k, m, int(nanmis.sum())))
            global worst
            if m > worst[0] and m < 1e30:
                worst = (m, path + "/" + k)


t0 = time.time()
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for it in range(n_cases):
        H, W = int(rng.integers(130, 900)), int(rng.integers(130, 900))
        if rng.random() < 0.3:
            W = H
        img = synth.speckle_frame(max(H, W), int(rng.integers(0, 1000)))[:H, :W].copy()
        kw = dict(tiles=bool(rng.random() < 0.7), display_origin=str(rng.choice(["lower", "upper"])))
        for name in ("speckle_stats", "sharpness_stats"):
            res = []
            for fn in (lambda: getattr(gm, name)(img, verbose=False, **kw), lambda: getattr(M, name)(img, **kw)):
                try:
                    res.append((fn(), None))
                except Exception as e:  # noqa: BLE001
                    res.append((None, type(e).__name__ + ": " + str(e)[:80]))
            if (res[0][1] is None) != (res[1][1] is None) or (res[0][1] and res[0][1].split(":")[0] != res[1][1].split(":")[0]):
                bad += 1; print("EXC-MISMATCH", name, (H, W), kw, res[0][1], "|", res[1][1], flush=True); continue
            if res[0][0] is None:
                continue
            out = []
            walk(res[0][0], res[1][0], name, out)
            if out:
                bad += 1; print("FAIL", name, (H, W), kw, out[:4], flush=True)
        print(f"  case {it} {(H, W)} {kw} ok so far, bad={bad}, {time.time() - t0:.0f} s", flush=True)
print(f"aggregator soak done: {n_cases} cases, {bad} failures, worst rel {worst[0]:.2e} at {worst[1]}, {time.time() - t0:.0f} s", flush=True)
