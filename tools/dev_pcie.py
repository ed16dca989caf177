"""dev: PCIe-inclusive rate of the headline path when the boundary is handed HOST arrays (NumPy in, NumPy out)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from barc4dip_amd import synth
from barc4dip_amd.signal import psd_autocorr2d_stack

T, n = 64, 2048
host = np.stack([synth.speckle_frame(n, 10 + i) for i in range(4)] * (T // 4))
psd_autocorr2d_stack(host[:4]); torch.cuda.synchronize()
for label, arr in (("pageable NumPy in/out", host),):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); p, a = psd_autocorr2d_stack(arr); best = min(best, time.perf_counter() - t0)
    gb = (arr.nbytes + p.nbytes + a.nbytes) / 1e9
    print(f"{label}: {T / best:.0f} frames/s ({gb / best:.1f} GB/s over PCIe, {gb:.1f} GB per call)", flush=True)
# pinned staging, asynchronous copies on the compute stream
pin_in = torch.from_numpy(host).pin_memory()
pin_p = torch.empty((T, n, n), dtype=torch.float32).pin_memory(); pin_a = torch.empty_like(pin_p).pin_memory()
best = 1e9
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dev = pin_in.to("cuda", non_blocking=True)
    p, a = psd_autocorr2d_stack(dev, return_tensors=True)
    pin_p.copy_(p, non_blocking=True); pin_a.copy_(a, non_blocking=True)
    torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
gb = 3 * host.nbytes / 1e9
print(f"pinned host buffers: {T / best:.0f} frames/s ({gb / best:.1f} GB/s over PCIe)", flush=True)
