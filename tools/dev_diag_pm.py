"""Phase timeline of the fused mixed-radix row kernel (cfg5 size) from the diagnostic build (developer tool, GPU box).
The build records the pass whose input mode is B4D_DIAG_PM_IN (default 3: the forward row-pair pass of the Wiener path)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi, synth  # noqa: E402

_ffi._lib = _ffi.load_library(sys.argv[2] if len(sys.argv) > 2 else "barc4dip_amd/csrc/libb4d_diag.so")
from barc4dip_amd.preprocessing import deconvolve_psf  # noqa: E402

lib = _ffi._lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = synth.speckle_stack_device(1, n)
deconvolve_psf(dev, sigma=1.5, return_tensors=True)
torch.cuda.synchronize()
rows = 4200
diag = torch.zeros((rows, 8), dtype=torch.int64, device="cuda")
lib.b4d_debug_set_pm_diag.argtypes = [C.c_void_p]
assert lib.b4d_debug_set_pm_diag(C.c_void_p(diag.data_ptr())) == 0
deconvolve_psf(dev, sigma=1.5, return_tensors=True)
torch.cuda.synchronize()
d = diag.cpu().numpy().astype(np.float64)
d = d[d[:, 0] > 0]
t0 = d[:, 0].min()
us = (d[:, :6] - t0) / 100.0          # wall_clock64: 100 MHz
if d[:, 6].max() > 0:
    print("inside load+radixP (us): inputs arrived %.2f, butterflies + twiddles + LDS writes %.2f, wait at the barrier %.2f" % (
        np.median(d[:, 6] - d[:, 1]) / 100, np.median(d[:, 7] - d[:, 6]) / 100, np.median(d[:, 2] - d[:, 7]) / 100))
names = ["tables", "load+radixP", "DFT_A", "DFT_B", "store issue"]
dt = np.diff(us, axis=1)
print("workgroups recorded:", len(d), " kernel span %.1f us" % (us[:, 5].max()))
print("phase medians (us):", {k: round(float(np.median(dt[:, i])), 2) for i, k in enumerate(names)})
print("phase means   (us):", {k: round(float(np.mean(dt[:, i])), 2) for i, k in enumerate(names)})
print("row total median %.2f us, mean %.2f" % (np.median(us[:, 5] - us[:, 0]), np.mean(us[:, 5] - us[:, 0])))
st = np.sort(us[:, 0])
print("start times (us) of workgroups #0, 255, 256, 511, 512, 767, 1023, 1024, last:", [round(float(st[min(i, len(st) - 1)]), 1) for i in (0, 255, 256, 511, 512, 767, 1023, 1024, len(st) - 1)])
print("resident at t = 5 us:", int(((us[:, 0] <= 5) & (us[:, 5] > 5)).sum()), " at mid-kernel:", int(((us[:, 0] <= us[:, 5].max() / 2) & (us[:, 5] > us[:, 5].max() / 2)).sum()))
