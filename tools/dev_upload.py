"""dev: host -> device time of _device.to_device_f32 for 8 x 2048^2 stacks of several dtypes (staged native-dtype route against the
plain astype + pageable copy)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _device as D  # noqa: E402

rng = np.random.default_rng(0)
for dtype in ("float32", "uint16", "float64"):
    a = (rng.random((8, 2048, 2048)) * 60000).astype(dtype)
    for route in ("staged", "plain"):
        D._UPLOAD_MIN_BYTES = (32 << 20) if route == "staged" else (1 << 60)
        D.to_device_f32(a, ndim=(3,))
        torch.cuda.synchronize()
        best = cold = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            D.to_device_f32(a, ndim=(3,))
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
            b = a.copy()                     # an array the runtime has not seen before
            t0 = time.perf_counter()
            D.to_device_f32(b, ndim=(3,))
            torch.cuda.synchronize()
            cold = min(cold, time.perf_counter() - t0)
            del b
        print(f"{dtype:8s} {route:6s}: same array {best * 1e3:6.1f} ms, fresh array {cold * 1e3:6.1f} ms  ({a.nbytes / cold / 1e9:5.1f} GB/s of host bytes)", flush=True)
