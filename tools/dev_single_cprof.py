"""dev: host-side profile (cProfile) of speckle_stats / sharpness_stats on one 2048^2 frame."""
import cProfile
import pstats
import sys
import warnings

import torch

sys.path.insert(0, ".")
from barc4dip_amd import metrics as gm, synth  # noqa: E402

warnings.simplefilter("ignore")
img = synth.speckle_frame(2048, 1234)
fn = gm.speckle_stats if (len(sys.argv) > 1 and sys.argv[1] == "speckle") else gm.sharpness_stats
fn(img, verbose=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
fn(img, verbose=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
