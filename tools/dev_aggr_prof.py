"""dev: speckle_stats / sharpness_stats at 2048^2 three times each (for rocprofv3 kernel statistics)."""
import sys
import time
import warnings

import torch

sys.path.insert(0, ".")
from barc4dip_amd import metrics as gm, synth  # noqa: E402

warnings.simplefilter("ignore")
img = synth.speckle_frame(2048, 1234)
for name, fn in (("speckle_stats", gm.speckle_stats), ("sharpness_stats", gm.sharpness_stats)):
    fn(img, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn(img, verbose=False)
    torch.cuda.synchronize()
    print(name, f"{(time.perf_counter() - t0) / 3 * 1e3:.1f} ms", flush=True)
