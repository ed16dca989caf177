"""dev: interleaved A/B of the FFT -> PSD -> autocorrelation pipeline at detector formats between two builds of libb4d."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi  # noqa: E402

libs = {k: _ffi.load_library(p) for k, p in (("A", sys.argv[1]), ("B", sys.argv[2]))}
for (ny, nx, T) in ((2160, 2560, 32), (1080, 1920, 128), (480, 640, 512)):
    st = torch.rand((T, ny, nx), device="cuda") * 1000
    psd, ac = torch.empty_like(st), torch.empty_like(st)
    plans = {}
    for k, lib in libs.items():
        h = C.c_void_p()
        assert lib.b4d_plan_create(ny, nx, _ffi.default_chunk(ny, nx), C.byref(h)) == 0, lib.b4d_last_error()
        plans[k] = h
    res = {"A": [], "B": []}
    outs = {}
    for rnd in range(8):
        for k, lib in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.b4d_psd_autocorr2d(plans[k], C.c_void_p(st.data_ptr()), T, C.c_void_p(psd.data_ptr()), 1.0, C.c_void_p(ac.data_ptr()), 3, None) == 0
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1))
            if rnd == 0:
                outs[k] = (psd.clone(), ac.clone())
    same = bool(torch.equal(outs["A"][0], outs["B"][0]) and torch.equal(outs["A"][1], outs["B"][1]))
    print(f"{ny}x{nx}: A {T / np.median(res['A'][2:]) * 1e3:.0f} frames/s, B {T / np.median(res['B'][2:]) * 1e3:.0f} frames/s, identical {same}", flush=True)
    for k, lib in libs.items():
        lib.b4d_plan_destroy(plans[k])
