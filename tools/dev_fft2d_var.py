"""dev: A/B of kernel variants behind b4d_set_option("exp", v) on fft2d_stack 2048^2 (interleaved, best of 5 each round)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi  # noqa: E402
from barc4dip_amd.signal.fft import fft2d_stack  # noqa: E402

lib = _ffi.lib()
variants = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]
ny = nx = 2048
T = 128
st = torch.rand((T, ny, nx), device="cuda") * 1000
ref = torch.fft.fftshift(torch.fft.fft2(st[:2].double()), dim=(-2, -1))
for rnd in range(3):
    for v in variants:
        assert lib.b4d_set_option(b"exp", v) == 0
        out = fft2d_stack(st, return_tensors=True)
        err = float((out[:2].to(torch.complex128) - ref).abs().max() / ref.abs().max())
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            fft2d_stack(st, return_tensors=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"round {rnd} variant {v}: {T / best:9.0f} frames/s  {12 * ny * nx * T / best / 8e12:.3f} of 8 TB/s (12 B/px)  err {err:.1e}", flush=True)
        del out
lib.b4d_set_option(b"exp", 0)
