"""dev: randomised soak of deconvolve_psf (wiener / rl) and flat_field_correction against the oracle."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from barc4dip_amd import preprocessing as gp, synth
from oracle import wiener_np as W, preprocess_np as P

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
t0 = time.time()
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for it in range(n_cases):
    H, W_ = int(rng.integers(40, 420)), int(rng.integers(40, 420))
    img = synth.speckle_frame(max(H, W_), int(rng.integers(0, 1000)))[:H, :W_].copy()
    sig = (float(rng.uniform(0.5, 2.2)), float(rng.uniform(0.5, 2.2))) if rng.random() < 0.5 else float(rng.uniform(0.5, 2.2))
    clip = bool(rng.random() < 0.5)
    for kw in (dict(sigma=sig, clip=clip), dict(sigma=sig, clip=clip, method="rl", num_iter=int(rng.integers(1, 8)))):
        try:
            a, b = gp.deconvolve_psf(img, **kw), W.deconvolve_psf(img, **kw)
            err = float(np.max(np.abs(a - b)) / np.max(np.abs(img)))
            if not err < 2e-5:
                bad += 1; print("FAIL deconv", (H, W_), kw, err, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1; print("EXC deconv", (H, W_), kw, repr(e)[:150], flush=True)
    T = int(rng.integers(1, 4))
    imgs = rng.integers(0, 4000, size=(T, H, W_)).astype(rng.choice([np.uint16, np.float32, np.int32]))
    flats = rng.integers(1500, 2500, size=(int(rng.integers(1, 4)), H, W_)).astype(np.uint16)
    darks = rng.integers(80, 120, size=(H, W_)).astype(np.uint16)
    flats[:, rng.integers(0, H, 5), rng.integers(0, W_, 5)] = 0
    kw = dict(flats=flats, darks=darks, scale=str(rng.choice(["flat_median", "none"])), bad_pixel_removal=bool(rng.random() < 0.5))
    try:
        a, b = gp.flat_field_correction(imgs, **kw), P.flat_field_correction(imgs, **kw)
        if not np.array_equal(a, b, equal_nan=True):
            bad += 1; print("FAIL flat", (T, H, W_), kw["scale"], kw["bad_pixel_removal"], int((a != b).sum()), flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1; print("EXC flat", (T, H, W_), repr(e)[:150], flush=True)
print(f"prep soak done: {n_cases} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
