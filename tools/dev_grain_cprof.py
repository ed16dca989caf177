"""dev: cProfile of the tile-grain block of speckle_stats (2048^2, 81 sub-tiles)."""
import cProfile
import pstats
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import synth  # noqa: E402
from barc4dip_amd.metrics import speckles as SP  # noqa: E402

warnings.simplefilter("ignore")
img = synth.speckle_frame(2048, 1234)
t = SP._dev2d(np.ascontiguousarray(img[::-1]))
for _ in range(3):
    SP.tiled_fields_batched(t, "subtiles_9x9", SP._grain_batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    SP.tiled_fields_batched(t, "subtiles_9x9", SP._grain_batch)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
