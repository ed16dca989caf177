"""Interleaved timing of several builds of libb4d in ONE process on the cfg2 workload (developer tool).

    python tools/dev_ab_multi.py [--chunk 256] [--rounds 8] libA.so libB.so ...

Per build: median per-kernel milliseconds per pass (row R2C, column, peak, row C2R: the library's own HIP events) and whether
PSD / autocorrelation of the first frames are bit-identical to the first build's."""
import argparse
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chunk", type=int, default=256)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--frames", type=int, default=256)
ap.add_argument("--n", type=int, default=2048)
ap.add_argument("--plans", type=int, default=1, help="repeat the comparison on this many plans (= workspace allocations)")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
T, n = a.frames, a.n
stack = synth.speckle_stack_device(T, n)
psd = torch.empty_like(stack)
ac = torch.empty_like(stack)
libs = [(p.split("/")[-1], _ffi.load_library(p)) for p in a.libs]
# ONE plan (created by the first build) serves every build: the plan struct is the same in all of them and the workspace
# address decides a few per cent of each kernel's time, which would drown the differences looked for.
# --plans K repeats the whole comparison on K workspace allocations (alive together), to see the builds in both placement classes.
hs = []
for _ in range(a.plans):
    h = C.c_void_p()
    assert libs[0][1].b4d_plan_create(n, n, a.chunk, C.byref(h)) == 0
    hs.append(h)
for pi, h in enumerate(hs):
    res = [[] for _ in libs]
    outs = [None] * len(libs)
    for rnd in range(a.rounds + 1):
        for i in [(k + rnd) % len(libs) for k in range(len(libs))]:   # rotate the order from round to round
            name, lib = libs[i]
            kms = (C.c_float * 4)()
            for _ in range(3):
                rc = lib.b4d_psd_autocorr2d_timed(h, C.c_void_p(stack.data_ptr()), T, C.c_void_p(psd.data_ptr()), 1.0,
                                                  C.c_void_p(ac.data_ptr()), 3, None, kms)
                assert rc == 0, (name, rc)
            torch.cuda.synchronize()
            res[i].append([v / 3 for v in kms])
            if rnd == 0:
                outs[i] = (psd[:3].clone(), ac[:3].clone(), psd[-1].clone(), ac[-1].clone())
    for i, (name, _) in enumerate(libs):
        r = np.array(res[i][1:])
        med = np.median(r, axis=0)
        same = all(bool(torch.equal(x, y)) for x, y in zip(outs[0], outs[i]))
        print("plan %d %-18s r2c %.3f col %.3f peak %.3f c2r %.3f total %.3f ms -> %.0f frames/s (min total %.3f)  identical to first: %s" %
              (pi, name, *med, med.sum(), T / med.sum() * 1e3, r.sum(axis=1).min(), same), flush=True)
