"""dev: randomised soak of the two trackers (sizes, ROI positions, odd / even templates) against the oracle."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from barc4dip_amd import signal as gs, synth
from oracle import signal_np as S, ncc_np as N

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
t0 = time.time()
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for it in range(n_cases):
    H, W = int(rng.integers(96, 700)), int(rng.integers(96, 700))
    if rng.random() < 0.3:
        H = W = int(2 ** rng.integers(7, 10))
    base = synth.speckle_frame(max(H, W), int(rng.integers(0, 1000)))[:H, :W]
    h, w = int(rng.integers(17, min(H, 200) - 8)), int(rng.integers(17, min(W, 200) - 8))
    y0, x0 = int(rng.integers(0, H - h + 1)), int(rng.integers(0, W - w + 1))
    dy, dx = int(rng.integers(-6, 7)), int(rng.integers(-6, 7))
    img = (np.roll(base, (dy, dx), axis=(0, 1)) + rng.normal(size=(H, W)) * 10).astype(np.float32)
    sl = (slice(y0, y0 + h), slice(x0, x0 + w))
    for name, gf, of, kw in (("phase", gs.phase_correlation, S.phase_correlation, {}),
                             ("ncc", gs.template_matching, N.template_matching, dict(backend="skimage"))):
        try:
            a = gf(base[sl], img, slices_yx=sl, subpixel=False, **kw)
            b = of(base[sl], img, slices_yx=sl, subpixel=False, **kw)
            a2 = gf(base[sl], img, slices_yx=sl, **kw)
            b2 = of(base[sl], img, slices_yx=sl, **kw)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("EXC", name, (H, W), (h, w), (y0, x0), repr(e)[:120], flush=True)
            continue
        ok = (a[0], a[1]) == (b[0], b[1]) and abs(a2[0] - b2[0]) < 2e-2 and abs(a2[1] - b2[1]) < 2e-2 and abs(a[2] - b[2]) <= 2e-3 * max(1.0, abs(b[2]))
        if not ok:
            bad += 1
            print("FAIL", name, (H, W), (h, w), (y0, x0), a, b, a2, b2, flush=True)
print(f"tracker soak done: {n_cases} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
