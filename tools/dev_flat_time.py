"""dev: flat_field_correction throughput on a resident stack (frame in, frame out: 8 bytes per pixel + the shared flat / dark)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from barc4dip_amd import synth  # noqa: E402
from barc4dip_amd.preprocessing import flat_field_correction  # noqa: E402

T, n = 256, 2048
dev = synth.speckle_stack_device(T, n)
flat = dev[:4].mean(dim=0) + 50.0
dark = torch.full((n, n), 3.0, device="cuda")
for scale in ("flat_median", "none"):
    flat_field_correction(dev[:8], flats=flat, darks=dark, scale=scale, return_tensors=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = flat_field_correction(dev, flats=flat, darks=dark, scale=scale, return_tensors=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"flat_field_correction scale={scale}: {T / dt:.0f} frames/s, {T * n * n * 8 / dt / 1e12:.2f} TB/s", flush=True)
