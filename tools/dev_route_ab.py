"""dev: pipeline throughput at sides whose fused split has a large factor, with a given library build."""
import sys, time
sys.path.insert(0, ".")
import torch
from barc4dip_amd import _ffi, synth
if len(sys.argv) > 1:
    _ffi._lib = _ffi.load_library(sys.argv[1])
from barc4dip_amd.signal.corr import psd_autocorr2d_stack
for n in (1370, 1758, 2055, 3014, 3507, 4014):
    T = 4
    dev = synth.speckle_stack_device(T, 4096)[:, :n, :n].contiguous()
    psd_autocorr2d_stack(dev[:1], return_tensors=True); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); psd_autocorr2d_stack(dev, return_tensors=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(sys.argv[1:] or "default", n, "frames/s %.0f" % (T / best), flush=True)
