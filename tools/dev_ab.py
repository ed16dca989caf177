"""Interleaved A/B timing of two builds of libb4d in ONE process (developer tool)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi, synth  # noqa: E402

T, n, chunk = 256, 2048, 64
stack = synth.speckle_stack_device(T, n)
psd = torch.empty_like(stack)
ac = torch.empty_like(stack)
libs = {name: _ffi.load_library(path) for name, path in (("A", sys.argv[1]), ("B", sys.argv[2]))}
plans = {}
for k, lib in libs.items():
    h = C.c_void_p()
    assert lib.b4d_plan_create(n, n, chunk, C.byref(h)) == 0
    plans[k] = h
res = {k: [] for k in libs}
outs = {}
for rnd in range(8):
    for k, lib in libs.items():
        kms = (C.c_float * 4)()
        for _ in range(3):
            rc = lib.b4d_psd_autocorr2d_timed(plans[k], C.c_void_p(stack.data_ptr()), T, C.c_void_p(psd.data_ptr()), 1.0,
                                              C.c_void_p(ac.data_ptr()), 3, None, kms)
            assert rc == 0
        res[k].append([v / 3 for v in kms])
        if rnd == 0:
            outs[k] = (psd[:2].clone(), ac[:2].clone())
for k in libs:
    a = np.array(res[k][1:])
    med = np.median(a, axis=0)
    print(k, sys.argv[1 if k == "A" else 2].split("/")[-1], "median ms: r2c %.3f col %.3f peak %.3f c2r %.3f total %.3f -> %.0f frames/s (min total %.3f)" %
          (*med, med.sum(), T / med.sum() * 1e3, a.sum(axis=1).min()))
print("outputs identical:", bool(torch.equal(outs["A"][0], outs["B"][0])), bool(torch.equal(outs["A"][1], outs["B"][1])),
      "max |dPSD| rel", float(((outs["A"][0] - outs["B"][0]).abs().max() / outs["A"][0].abs().max())))
