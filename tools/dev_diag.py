"""Phase timeline of the column kernel from the diagnostic build (developer tool; run on the GPU box)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi, synth  # noqa: E402

lib = _ffi.load_library("barc4dip_amd/csrc/libb4d_diag.so")
T = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
n, chunk = 2048, 64
stack = synth.speckle_stack_device(T, n)
psd = torch.empty_like(stack)
ac = torch.empty_like(stack)
h = C.c_void_p()
assert lib.b4d_plan_create(n, n, chunk, C.byref(h)) == 0
nt = n // 2 // 16
diag = torch.zeros((T * nt, 8), dtype=torch.int64, device="cuda")
lib.b4d_debug_set_diag.argtypes = [C.c_void_p]
for it in range(2):
    lib.b4d_debug_set_diag(C.c_void_p(diag.data_ptr()))
    rc = lib.b4d_psd_autocorr2d(h, C.c_void_p(stack.data_ptr()), T, C.c_void_p(psd.data_ptr() if "nopsd" not in sys.argv else 0),
                                1.0, C.c_void_p(ac.data_ptr()), 3, None)
    assert rc == 0
    torch.cuda.synchronize()
d = diag.cpu().numpy().astype(np.float64)
names = ["load", "fwd", "stage3+psd", "psd drain", "inverse", "pairing+store issue", "store drain"]
dt = np.diff(d, axis=1) / 1000.0  # s_memtime ticks = shader cycles -> kilo-cycles
print("phase medians (kcycles):", {k: round(float(np.median(dt[:, i])), 2) for i, k in enumerate(names)})
print("tile total median %.1f kcyc, mean %.1f; kernel span %.0f kcyc for %d tiles" %
      (np.median(d[:, 7] - d[:, 0]) / 1000, np.mean(d[:, 7] - d[:, 0]) / 1000, (d[:, 7].max() - d[d[:, 0] > 0, 0].min()) / 1000, len(d)))
