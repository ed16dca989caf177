"""dev: wall-clock of the blocks of speckle_stats on one 2048^2 frame (device synchronised around each block)."""
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import synth  # noqa: E402
from barc4dip_amd.metrics import speckles as SP  # noqa: E402
from barc4dip_amd.metrics.common import choose_tiling_mode  # noqa: E402

warnings.simplefilter("ignore")
img = synth.speckle_frame(2048, 1234)
SP.speckle_stats(img, verbose=False)
torch.cuda.synchronize()


def timed(name, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / reps * 1e3:7.2f} ms", flush=True)
    return out


flipped = np.ascontiguousarray(img[::-1])
t = timed("upload (_dev2d)", lambda: SP._dev2d(flipped))
timed("amplitude (full)", lambda: SP.amplitude(t))
timed("grain (full, incl. autocorr D2H)", lambda: SP.grain(t))
from barc4dip_amd.metrics.statistics import distribution_moments  # noqa: E402

timed("stats (full)", lambda: distribution_moments(t, saturation_value=65535.0, eps=1e-6, verbose=False))
timed("bandwidth (full)", lambda: SP.bandwidth(t))
mode, _ = choose_tiling_mode(2048, 2048, tiles=True, min_tile_px=128)
timed("tiles pointwise", lambda: SP._tiles_pointwise(t, mode, True, True, 65535.0, 1e-6))
timed("tiles grain", lambda: SP.tiled_fields_batched(t, mode, SP._grain_batch))
timed("tiles bandwidth", lambda: SP.tiled_fields_batched(t, mode, SP._bandwidth_batch))
timed("speckle_stats (whole)", lambda: SP.speckle_stats(img, verbose=False))
