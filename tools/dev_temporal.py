"""dev: kernel-only rate of b4d_temporal_accumulate (cfg4 shard shape)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi  # noqa: E402

lib = _ffi.load_library(sys.argv[1]) if len(sys.argv) > 1 else _ffi.lib()
T, n = 256, 2048
stack = torch.rand((T, n, n), device="cuda") * 1000
sx = torch.zeros((n, n), dtype=torch.float64, device="cuda")
sxx = torch.zeros_like(sx)
npix = n * n
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        rc = lib.b4d_temporal_accumulate(C.c_void_p(stack.data_ptr()), T, npix, C.c_void_p(sx.data_ptr()), C.c_void_p(sxx.data_ptr()),
                                         _ffi.stream_ptr())
        assert rc == 0
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    print(f"temporal_accumulate: {ms:.3f} ms per {T} frames -> {T * npix * 4 / ms / 1e9:.2f} TB/s", flush=True)
ref = stack.double().sum(0) * 30
print("sum check rel err", float(((sx - ref).abs().max() / ref.abs().max())))
