"""dev: cfg5 throughput with a given library build."""
import sys, time
sys.path.insert(0, ".")
import torch
from barc4dip_amd import _ffi, synth
if len(sys.argv) > 1:
    _ffi._lib = _ffi.load_library(sys.argv[1])
from barc4dip_amd.preprocessing import deconvolve_psf
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = synth.speckle_stack_device(T, 4096)
deconvolve_psf(dev[:1], sigma=1.5, return_tensors=True); torch.cuda.synchronize()
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); deconvolve_psf(dev, sigma=1.5, return_tensors=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(sys.argv[1:] or "default", "cfg5 frames/s %.0f" % (T / best))
