"""dev: interleaved A/B of b4d_temporal_accumulate between two builds of libb4d in ONE process (cfg4 shard: 1024 x 2048^2)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi, synth  # noqa: E402

T, n = 512, 2048
dev = synth.speckle_stack_device(T, n)
libs = {k: _ffi.load_library(p) for k, p in (("A", sys.argv[1]), ("B", sys.argv[2]))}
sx = torch.zeros(n * n, dtype=torch.float64, device="cuda")
sxx = torch.zeros_like(sx)
res = {"A": [], "B": []}
for rnd in range(10):
    for k, lib in libs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2):
            assert lib.b4d_temporal_accumulate(C.c_void_p(dev.data_ptr()), T, n * n, C.c_void_p(sx.data_ptr()), C.c_void_p(sxx.data_ptr()), None) == 0
        e1.record()
        torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1) / 2)
for k in res:
    ms = np.median(res[k][2:])
    print(k, sys.argv[1 if k == "A" else 2].split("/")[-1], f"median {ms:.3f} ms -> {T * n * n * 4 / ms / 1e9:.2f} TB/s (min {min(res[k]):.3f} ms)")
