"""dev: streamed temporal statistics from host memory (cfg4 shape, uint16 source) -- PCIe-inclusive rate."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from barc4dip_amd import ingest
T, n = 256, 2048
rng = np.random.default_rng(0)
host = rng.integers(0, 4000, size=(T, n, n), dtype=np.uint16)
ingest.temporal_stats_streamed(host[:32], chunk_frames=16); torch.cuda.synchronize()
for chunk in (16, 32, 64):
    t0 = time.perf_counter(); ingest.temporal_stats_streamed(host, chunk_frames=chunk, return_tensors=True); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"chunk {chunk}: {T / dt:.0f} frames/s ({T * n * n * 4 / dt / 1e9:.1f} GB/s float32 over PCIe, host uint16->float32 conversion included)", flush=True)
