"""dev: deconvolve_psf (Wiener, sigma 1.5) frames/s at several frame sizes (B4D_WIENER_LANES2=1 for the one-lane driver)."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from barc4dip_amd import synth  # noqa: E402
from barc4dip_amd.preprocessing import deconvolve_psf  # noqa: E402

CASES = ((512, 512), (1024, 256), (2048, 64), (4096, 32), (4096, 64), (4096, 8))
sel = [int(a) for a in sys.argv[1:]]
for n, T in (CASES if not sel else [CASES[i] for i in sel]):
    dev = synth.speckle_stack_device(T, n)
    deconvolve_psf(dev[:2], sigma=1.5, return_tensors=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        deconvolve_psf(dev, sigma=1.5, return_tensors=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"lanes {os.environ.get('B4D_WIENER_LANES2', 'default')}: {n}^2 x {T}: {T / best:9.0f} frames/s", flush=True)
    del dev
    torch.cuda.empty_cache()
