"""dev: interleaved A/B of kernel variants behind b4d_set_option("exp", v) in ONE process (same tensors, same plans).
usage: dev_ab_exp.py <workload> <ny> <nx> <T> [variants...]     workload: fft2d | pipe (psd + autocorr) | wiener | track"""
import sys
import time

import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi  # noqa: E402
from barc4dip_amd.signal import psd_autocorr2d_stack  # noqa: E402
from barc4dip_amd.signal.fft import fft2d_stack  # noqa: E402

lib = _ffi.lib()
work, ny, nx, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
variants = [int(a) for a in sys.argv[5:]] or [0, 1]
st = torch.rand((T, ny, nx), device="cuda") * 1000
if work == "fft2d":
    run = lambda: fft2d_stack(st, return_tensors=True)  # noqa: E731
elif work == "pipe":
    run = lambda: psd_autocorr2d_stack(st, return_tensors=True)  # noqa: E731
elif work == "wiener":
    from barc4dip_amd.preprocessing import deconvolve_psf

    run = lambda: deconvolve_psf(st, sigma=1.5, return_tensors=True)  # noqa: E731
else:
    raise SystemExit("unknown workload")
ref = None
for rnd in range(3):
    for v in variants:
        assert lib.b4d_set_option(b"exp", v) == 0
        out = run()
        out = out if isinstance(out, torch.Tensor) else out[-1]
        if ref is None:
            ref = out[:2].clone()
        same = bool(torch.equal(out[:2], ref))
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{work} {ny}x{nx} round {rnd} variant {v}: {T / best:9.0f} frames/s  ({best / T * 1e6:.2f} us/frame)  identical to variant {variants[0]}: {same}", flush=True)
        del out
lib.b4d_set_option(b"exp", 0)
