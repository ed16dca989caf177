import sys
sys.path.insert(0, ".")
import numpy as np
from barc4dip_amd import synth
from barc4dip_amd import signal as gs
from oracle import signal_np as S
for shape in ((720, 600), (600, 720), (720, 1280)):
    H, W = shape
    rng = np.random.default_rng(H + W)
    base = synth.speckle_frame(max(H, W), 5)[:H, :W]
    for rep in range(2):
        for (dy, dx), (h, w) in (((3, -5), (61, 61)), ((-7, 11), (41, 81)), ((-7, 11), (61, 61)), ((3, -5), (41, 81))):
            fr = (np.roll(base, (dy, dx), axis=(0, 1)) + rng.normal(size=(H, W)) * 20).astype(np.float32)
            y0, x0 = (H - h) // 2 - 10, (W - w) // 2 + 7
            sl = (slice(y0, y0 + h), slice(x0, x0 + w))
            gi = gs.phase_correlation(base[sl], fr, slices_yx=sl, subpixel=False)
            wi = S.phase_correlation(base[sl], fr, slices_yx=sl, subpixel=False)
            print(shape, rep, (dy, dx), (h, w), "got", gi[:2], "want", wi[:2], "peak", round(gi[2], 4), round(wi[2], 4), flush=True)
