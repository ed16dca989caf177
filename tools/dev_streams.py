"""dev: does running small chunks on two streams (tail filling + infinity-cache reuse of the spectrum) beat one big chunk?"""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import torch
from barc4dip_amd import _ffi, synth
T, n = 256, 2048
stack = synth.speckle_stack_device(T, n)
psd = torch.empty_like(stack); ac = torch.empty_like(stack)
lib = _ffi.lib()
def run(chunk, nstreams, steps=5):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    plans = []
    for s in streams:
        h = C.c_void_p(); assert lib.b4d_plan_create(n, n, chunk, C.byref(h)) == 0; plans.append(h)
    def once():
        g = 0
        for a in range(0, T, chunk):
            s = streams[g % nstreams]; pl = plans[g % nstreams]; g += 1
            rc = lib.b4d_psd_autocorr2d(pl, C.c_void_p(stack[a].data_ptr()), min(chunk, T - a), C.c_void_p(psd[a].data_ptr()), 1.0 / (n * n),
                                        C.c_void_p(ac[a].data_ptr()), 3, C.c_void_p(s.cuda_stream))
            assert rc == 0
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for h in plans: lib.b4d_plan_destroy(h)
    print(f"chunk {chunk:3d} x {nstreams} stream(s): {T / dt:.0f} frames/s", flush=True)
for chunk, ns in ((64, 1), (32, 2), (16, 2), (8, 2), (8, 4), (4, 4), (16, 4)):
    run(chunk, ns)
