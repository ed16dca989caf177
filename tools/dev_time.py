"""Per-kernel timing experiments on the GPU box (developer tool)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi, synth  # noqa: E402


LIBPATH = None


def run(T, n, chunk, psd_on=True, ac_on=True, steps=5, flags=3):
    stack = synth.speckle_stack_device(T, n)
    psd = torch.empty_like(stack) if psd_on else None
    ac = torch.empty_like(stack) if ac_on else None
    lib = _ffi.load_library(LIBPATH) if LIBPATH else _ffi.lib()
    h = C.c_void_p()
    assert lib.b4d_plan_create(n, n, chunk, C.byref(h)) == 0

    class pl:
        handle = h

        @staticmethod
        def close():
            lib.b4d_plan_destroy(h)
    kms = (C.c_float * 4)()
    args = (pl.handle, C.c_void_p(stack.data_ptr()), T, C.c_void_p(psd.data_ptr() if psd_on else 0), 1.0 / (n * n),
            C.c_void_p(ac.data_ptr() if ac_on else 0), flags, _ffi.stream_ptr())
    for _ in range(2):
        _ffi.check(lib.b4d_psd_autocorr2d(*args))
    torch.cuda.synchronize()
    for _ in range(steps):
        _ffi.check(lib.b4d_psd_autocorr2d_timed(*args, kms))
    k = [v / steps for v in kms]
    tot = sum(k)
    print(f"T={T} n={n} chunk={chunk} psd={psd_on} ac={ac_on}: r2c {k[0]:.3f} col {k[1]:.3f} peak {k[2]:.3f} c2r {k[3]:.3f} "
          f"total {tot:.3f} ms -> {T/tot*1e3:.0f} frames/s; col per frame {k[1]/T*1e3:.2f} us", flush=True)
    pl.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        LIBPATH = sys.argv[1]
        run(256, 2048, 64)
        run(256, 2048, 64, psd_on=False)
        sys.exit(0)
    run(256, 2048, 32)
    run(256, 2048, 32, psd_on=False)
    run(256, 2048, 32, ac_on=False)
    run(256, 2048, 64)
    run(256, 2048, 16)
    run(512, 1024, 64)
