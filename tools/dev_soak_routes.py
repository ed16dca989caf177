"""dev: the three median routes of b4d_phase_correlation (expectation on / off / deliberately wrong) on awkward power-of-two
inputs -- sparse frames, constant regions, saturated blocks, tiny templates -- must agree bit for bit."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from barc4dip_amd import _ffi, signal as gs, synth  # noqa: E402

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
lib = _ffi.lib()
bad = 0
fallbacks = 0
t0 = time.time()
for it in range(n_cases):
    H, W = (int(2 ** rng.integers(6, 11)) for _ in range(2))
    kind = int(rng.integers(0, 5))
    base = synth.speckle_frame(max(H, W), int(rng.integers(0, 1000)))[:H, :W].copy()
    if kind == 1:
        base[rng.random((H, W)) < 0.97] = 0.0                       # sparse
    elif kind == 2:
        base[: H // 2] = 5.0                                        # half the frame constant
    elif kind == 3:
        base = np.minimum(base, np.percentile(base, 40)).astype(np.float32)   # saturated
    elif kind == 4:
        base = (rng.integers(0, 3, size=(H, W)) * 100).astype(np.float32)     # three grey levels
    T = 3
    stack = np.stack([np.roll(base, (t, -2 * t), axis=(0, 1)) for t in range(T)]).astype(np.float32)
    h, w = int(rng.integers(5, max(6, H // 3))), int(rng.integers(5, max(6, W // 3)))
    y0, x0 = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
    rois = [(y0, y0 + h, x0, x0 + w)]
    args = (stack, stack, [0], rois, list(range(T)), [0] * T)
    out = {}
    try:
        for mode in (1, 0, 2):
            assert lib.b4d_set_option(b"track_predict_bin", mode) == 0
            out[mode] = gs.phase_correlation_batch(*args, return_peak_ij=True)
    finally:
        lib.b4d_set_option(b"track_predict_bin", 1)
    same = all(np.array_equal(out[m][0], out[0][0], equal_nan=True) and np.array_equal(out[m][1], out[0][1]) for m in (1, 2))
    if not same:
        bad += 1
        print("MISMATCH", (H, W), kind, rois, out[1][0], out[0][0], out[2][0], flush=True)
print(f"route soak done: {n_cases} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
