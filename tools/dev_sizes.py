"""dev: pipeline throughput across plan sizes (power-of-two FFT kernels, DFT-matrix and fused mixed-radix plans)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi, synth  # noqa: E402
from barc4dip_amd.signal import psd_autocorr2d_stack  # noqa: E402

for (ny, nx, T) in ((512, 512, 1024), (1024, 1024, 512), (2048, 2048, 256), (4096, 4096, 64), (228, 228, 2048), (2160, 2560, 64), (4104, 4104, 16),
                    (1080, 1920, 128), (1200, 1600, 128), (1536, 2048, 64), (2048, 2448, 64), (3000, 4096, 16), (480, 640, 512)):
    st = torch.rand((T, ny, nx), device="cuda") * 1000
    psd_autocorr2d_stack(st, return_tensors=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        psd_autocorr2d_stack(st, return_tensors=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    nh = nx // 2 + 1
    bpipe = 12 * ny * nx + 40 * ny * nh
    print(f"{ny}x{nx}: {T / dt:9.0f} frames/s  {T * ny * nx / dt / 1e9:7.1f} Gpx/s  B_pipe rate {bpipe * T / dt / 1e12:5.2f} TB/s  chunk {_ffi.default_chunk(ny, nx)}", flush=True)
    del st
