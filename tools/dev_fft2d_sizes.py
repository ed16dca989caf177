"""dev: fft2d_stack (full fftshift-ed complex spectrum) throughput at general sizes + check against torch.fft on the device."""
import sys
import time

import torch

sys.path.insert(0, ".")
from barc4dip_amd.signal.fft import fft2d_stack  # noqa: E402

for (ny, nx, T) in ((600, 600, 256), (720, 1280, 128), (2160, 2560, 32), (264, 520, 256), (4104, 4104, 8), (1000, 2048, 32), (1080, 1920, 64), (2048, 2448, 32), (3000, 4096, 8)):
    st = torch.rand((T, ny, nx), device="cuda") * 1000
    out = fft2d_stack(st, return_tensors=True)
    ref = torch.fft.fftshift(torch.fft.fft2(st.double()), dim=(-2, -1))
    err = float((out.to(torch.complex128) - ref).abs().max() / ref.abs().max())
    del ref
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fft2d_stack(st, return_tensors=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"{ny}x{nx}: {T / dt:9.0f} frames/s  {12 * ny * nx * T / dt / 1e12:5.2f} TB/s (4 B in + 8 B out per px)  err {err:.2e}", flush=True)
    del st, out
