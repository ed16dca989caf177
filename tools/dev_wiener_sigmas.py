"""dev: deconvolve_psf throughput on 4096^2 frames over PSF widths (padded sizes with awkward factors)."""
import sys, time
sys.path.insert(0, ".")
import torch
from barc4dip_amd import synth
from barc4dip_amd.preprocessing import deconvolve_psf

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = synth.speckle_stack_device(4, n)
for sigma in (0.7, 1.0, 1.5, 2.0, 2.5, 3.0):
    try:
        t0 = time.perf_counter(); deconvolve_psf(dev[:1], sigma=sigma, return_tensors=True); torch.cuda.synchronize(); t_first = time.perf_counter() - t0
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); deconvolve_psf(dev, sigma=sigma, return_tensors=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"sigma {sigma}: first call {t_first:.2f} s, {4 / best:.1f} frames/s", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"sigma {sigma}: {type(e).__name__}: {str(e)[:120]}", flush=True)
