"""dev: host-side profile (cProfile) of speckle_stack_stats on an 8-frame 2048^2 stack."""
import cProfile
import pstats
import sys
import warnings

import torch

sys.path.insert(0, ".")
from barc4dip_amd import metrics as gm, synth  # noqa: E402

warnings.simplefilter("ignore")
stack, _ = synth.shifted_stack(8, 2048, seed=3, max_shift=16)
which = sys.argv[1] if len(sys.argv) > 1 else "speckle"
fn = gm.speckle_stack_stats if which == "speckle" else gm.sharpness_stack_stats
fn(stack, verbose=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
fn(stack, verbose=False)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
