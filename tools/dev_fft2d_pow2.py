"""dev: fft2d_stack at power-of-two sizes and two detector formats: frames/s, TB/s on the 12 B / pixel model, error vs torch.fft
(float64)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi  # noqa: E402
from barc4dip_amd.signal.fft import fft2d_stack  # noqa: E402

sizes = ((2048, 2048, 128), (1024, 1024, 256), (4096, 4096, 32), (512, 512, 512), (2160, 2560, 64), (1024, 2048, 128), (2048, 1024, 128),
         (256, 256, 512), (64, 64, 512), (64, 4096, 64), (4096, 64, 64), (128, 512, 256))
if len(sys.argv) > 1:
    sizes = sizes[: int(sys.argv[1])]
lib = _ffi.lib()
for (ny, nx, T) in sizes:
    pow2 = (ny & (ny - 1)) == 0 and (nx & (nx - 1)) == 0
    st = torch.rand((T, ny, nx), device="cuda") * 1000
    for route in (0,):
        out = fft2d_stack(st, return_tensors=True)
        err = 0.0
        for sl in (slice(0, 3), slice(T - 2, T)):
            ref = torch.fft.fftshift(torch.fft.fft2(st[sl].double()), dim=(-2, -1))
            err = max(err, float((out[sl].to(torch.complex128) - ref).abs().max() / ref.abs().max()))
            del ref
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            fft2d_stack(st, return_tensors=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{ny}x{nx} route {route}: {T / best:9.0f} frames/s  {12 * ny * nx * T / best / 1e12:5.2f} TB/s model (12 B/px) = "
              f"{12 * ny * nx * T / best / 8e12:.3f} of 8 TB/s  err {err:.2e}", flush=True)
        del out
    del st
    torch.cuda.empty_cache()
