"""dev: wall clock of speckle_stack_stats / sharpness_stack_stats per frame on host stacks (float32 and uint16 detector words)."""
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import metrics as gm, synth  # noqa: E402

warnings.simplefilter("ignore")
for T in (8, 32):
    stack, _ = synth.shifted_stack(T, 2048, seed=3, max_shift=16)
    for dtype in (np.float32, np.uint16):
        st = np.clip(stack, 0, 65535).astype(dtype)
        for name, fn in (("speckle_stack_stats", gm.speckle_stack_stats), ("sharpness_stack_stats", gm.sharpness_stack_stats)):
            fn(st, verbose=False)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                fn(st, verbose=False)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            print(f"T={T:3d} {np.dtype(dtype).name:8s} {name:22s}: {best * 1e3:7.1f} ms = {best / T * 1e3:5.2f} ms per frame", flush=True)
