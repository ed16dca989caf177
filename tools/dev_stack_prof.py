"""dev: speckle_stack_stats / sharpness_stack_stats on an 8-frame 2048^2 stack (wall clock; for rocprofv3 kernel statistics)."""
import sys
import time
import warnings

import torch

sys.path.insert(0, ".")
from barc4dip_amd import metrics as gm, synth  # noqa: E402

warnings.simplefilter("ignore")
stack, _ = synth.shifted_stack(8, 2048, seed=3, max_shift=16)
for name, fn in (("speckle_stack_stats", gm.speckle_stack_stats), ("sharpness_stack_stats", gm.sharpness_stack_stats)):
    fn(stack, verbose=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(stack, verbose=False)
    torch.cuda.synchronize()
    print(name, f"{(time.perf_counter() - t0) / 8 * 1e3:.1f} ms per frame", flush=True)
