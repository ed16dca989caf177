"""dev: psd + autocorrelation pipeline against the frames per launch group, several frame sizes (C ABI, events)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi  # noqa: E402

lib = _ffi.lib()
for n, T in ((512, 4096), (1024, 1024), (4096, 64)):
    st = torch.rand((T, n, n), device="cuda") * 1000
    psd, ac = torch.empty_like(st), torch.empty_like(st)
    base = _ffi.default_chunk(n, n)
    row = []
    for chunk in (base, 2 * base, 4 * base, base, 4 * base):
        h = C.c_void_p()
        assert lib.b4d_plan_create(n, n, chunk, C.byref(h)) == 0
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.b4d_psd_autocorr2d(h, C.c_void_p(st.data_ptr()), T, C.c_void_p(psd.data_ptr()), 1.0, C.c_void_p(ac.data_ptr()), 3, None) == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        lib.b4d_plan_destroy(h)
        row.append(f"chunk {chunk}: {T / np.median(ts[2:]) * 1e3:.0f}")
    print(f"{n}^2 x {T}: " + " | ".join(row) + " frames/s", flush=True)
    del st, psd, ac
