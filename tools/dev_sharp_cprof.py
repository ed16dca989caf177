"""dev: host-side profile of sharpness_stats / speckle_stats at 2048^2 with tiles (cProfile, cumulative)."""
import cProfile
import pstats
import sys
import warnings

import torch

sys.path.insert(0, ".")
from barc4dip_amd import metrics as gm, synth  # noqa: E402

warnings.simplefilter("ignore")
img = synth.speckle_frame(2048, 1234)
for fn in (gm.sharpness_stats, gm.speckle_stats):
    fn(img, verbose=False)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        fn(img, verbose=False)
    torch.cuda.synchronize()
    pr.disable()
    print("=====", fn.__name__)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
