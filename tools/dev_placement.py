"""How much of a kernel's time is decided by WHERE its buffers were allocated (developer tool).

Same library, same kernels, same input.  (1) NP plans per round (each with its own hipMalloc'ed workspace), timed round-robin
on the cfg2 workload, destroyed, next round; (2) the best plan against output pairs (PSD, autocorrelation) from plain hipMalloc
and from hipExtMallocWithFlags(hipDeviceMallocContiguous); (3) the input stack either way.  Findings: DESIGN.md §8.6."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi, synth  # noqa: E402

T, n, chunk, NP = 256, 2048, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 6
lib = _ffi.lib()
hip = C.CDLL("libamdhip64.so")
stack = synth.speckle_stack_device(T, n)
psd = torch.empty_like(stack)
ac = torch.empty_like(stack)
NBYTES = stack.numel() * 4


def dev_alloc(contig):
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(NBYTES), C.c_uint(4)) if contig else hip.hipMalloc(C.byref(p), C.c_size_t(NBYTES))
    assert rc == 0, rc
    return p.value


def time_sets(sets, rounds=4):
    """sets: list of (plan, stack_ptr, psd_ptr, ac_ptr)"""
    res = [[] for _ in sets]
    for rnd in range(rounds + 1):
        for k in range(len(sets)):
            i = (k + rnd) % len(sets)
            kms = (C.c_float * 4)()
            pl, s_, p_, a_ = sets[i]
            for _ in range(3):
                assert lib.b4d_psd_autocorr2d_timed(pl, C.c_void_p(s_), T, C.c_void_p(p_), 1.0, C.c_void_p(a_), 3, None, kms) == 0
            torch.cuda.synchronize()
            res[i].append([v / 3 for v in kms])
    return [np.median(np.array(r[1:]), axis=0) for r in res]


def show(tag, meds, extra=None):
    for i, m in enumerate(meds):
        print("%s %d: r2c %.3f col %.3f c2r %.3f total %.3f ms%s" % (tag, i, m[0], m[1], m[3], m.sum(), extra[i] if extra else ""), flush=True)


S0, P0, A0 = stack.data_ptr(), psd.data_ptr(), ac.data_ptr()
best = None
for rep in range(3):
    plans = []
    for _ in range(NP):
        h = C.c_void_p()
        assert lib.b4d_plan_create(n, n, chunk, C.byref(h)) == 0
        plans.append(h)
    meds = time_sets([(h, S0, P0, A0) for h in plans])
    show("allocation round %d, plan" % rep, meds)
    if rep < 2:
        for h in plans:
            lib.b4d_plan_destroy(h)
bp = plans[int(np.argmin([m.sum() for m in meds]))]
for contig in (False, True, False, True):
    bufs = [(dev_alloc(contig), dev_alloc(contig)) for _ in range(4)]
    meds = time_sets([(bp, S0, p_, a_) for p_, a_ in bufs])
    show("best plan, %s output pair" % ("contiguous" if contig else "hipMalloc"), meds)
    for p_, a_ in bufs:
        hip.hipFree(C.c_void_p(p_))
        hip.hipFree(C.c_void_p(a_))
for contig in (False, True):
    ins = [dev_alloc(contig) for _ in range(4)]
    for q in ins:
        assert hip.hipMemcpy(C.c_void_p(q), C.c_void_p(S0), C.c_size_t(NBYTES), C.c_int(3)) == 0
    meds = time_sets([(bp, q, P0, A0) for q in ins])
    show("best plan, %s input stack" % ("contiguous" if contig else "hipMalloc"), meds)
    for q in ins:
        hip.hipFree(C.c_void_p(q))
