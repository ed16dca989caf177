"""dev: cfg5 throughput (4096^2, sigma 1.5) with a given library build / frames-per-launch; error vs the oracle on frame 0."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from barc4dip_amd import _ffi, synth
if len(sys.argv) > 1 and sys.argv[1] != "-":
    _ffi._lib = _ffi.load_library(sys.argv[1])
from barc4dip_amd.preprocessing import deconvolve_psf
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = synth.speckle_stack_device(T, 4096)
out = deconvolve_psf(dev[:1], sigma=1.5, return_tensors=True); torch.cuda.synchronize()
if "check" in sys.argv:
    from oracle import wiener_np as W
    h = dev[0].cpu().numpy()
    ref = W.deconvolve_psf(h, sigma=1.5)
    print("err/range %.2e" % (float(np.max(np.abs(out[0].cpu().numpy() - ref))) / float(h.max())))
best = 1e9
for _ in range(5):
    t0 = time.perf_counter(); deconvolve_psf(dev, sigma=1.5, return_tensors=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
import os
print(sys.argv[1:] or "default", "FPL", os.environ.get("B4D_WIENER_FPL"), "cfg5 frames/s %.0f  (%.1f us/frame)" % (T / best, best / T * 1e6))
