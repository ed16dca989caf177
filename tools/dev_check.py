"""Developer smoke/timing script for the GPU box (not part of the product or the test suite)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import signal as gs, synth, _ffi  # noqa: E402
from oracle import signal_np as S  # noqa: E402


def relerr(got, ref):
    return float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))


def check(ny, nx):
    n = max(ny, nx)
    img = synth.speckle_frame(n, 1234)[:ny, :nx].copy()
    img64 = img.astype(np.float64)
    F, _, _ = gs.fft2d(img)
    Fr = S.fft2d(img64)[0]
    P, _, _ = gs.psd2d(img)
    Pr = S.psd2d(img64)[0]
    ac, _, _ = gs.autocorr2d(img)
    acr = S.autocorr2d(img64)[0]
    Pnd = P.copy(); Pnd[ny // 2, nx // 2] = 0
    Prnd = Pr.copy(); Prnd[ny // 2, nx // 2] = 0
    print(f"{ny}x{nx}: fft2d {relerr(F, Fr):.2e}  psd {relerr(P, Pr):.2e} (noDC l2 {np.linalg.norm(Pnd-Prnd)/np.linalg.norm(Prnd):.2e})"
          f"  autocorr maxabs {np.max(np.abs(ac-acr)):.2e} peak@{np.unravel_index(np.argmax(ac), ac.shape)}={ac.max()!r}", flush=True)


def timeit(T, n, chunk, iters=3):
    stack = synth.speckle_stack_device(T, n)
    psd = torch.empty_like(stack)
    ac = torch.empty_like(stack)
    pl = _ffi.Plan(n, n, chunk)
    lib = _ffi.lib()
    import ctypes as C
    args = (pl.handle, C.c_void_p(stack.data_ptr()), T, C.c_void_p(psd.data_ptr()), 1.0 / (n * n),
            C.c_void_p(ac.data_ptr()), 3, _ffi.stream_ptr())
    _ffi.check(lib.b4d_psd_autocorr2d(*args))
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(iters):
        t0 = time.perf_counter()
        _ffi.check(lib.b4d_psd_autocorr2d(*args))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    bpipe = 12 * n * n + 40 * n * (n // 2 + 1)
    print(f"T={T} n={n} chunk={chunk}: {best*1e3:.2f} ms  {T/best:.0f} frames/s  {bpipe*T/best/1e12:.2f} TB/s algorithmic", flush=True)
    pl.close()


if __name__ == "__main__":
    for ny, nx in ((512, 512), (1024, 1024), (2048, 2048), (512, 1024), (2048, 1024), (4096, 4096)):
        check(ny, nx)
    for chunk in (2, 4, 8, 16, 32):
        timeit(64, 2048, chunk)
    timeit(256, 1024, 32)
