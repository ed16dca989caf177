"""dev: device -> host time of _device.to_host for large results (float32 device tensor -> float64 / float32 arrays)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _device as D  # noqa: E402

for T in (8, 16, 32):
    t = torch.rand((T, 2048, 2048), device="cuda")
    for want in (np.float64, np.float32):
        for route in ("staged", "plain"):
            D._PINNED_MAX_BYTES = (1 << 30) if route == "staged" else (1 << 60)
            if route == "plain":
                f = lambda: t.cpu().numpy().astype(want, copy=False)  # noqa: E731  (what the plain route does beyond 256 MB)
            else:
                f = lambda: D.to_host(t, want)  # noqa: E731
            r = f()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                r = f()
                best = min(best, time.perf_counter() - t0)
            ok = bool(np.array_equal(r, t.cpu().numpy().astype(want)))
            print(f"T={T:3d} -> {np.dtype(want).name:8s} {route:6s}: {best * 1e3:7.1f} ms  ({r.nbytes / best / 1e9:5.1f} GB/s of result bytes) equal {ok}", flush=True)
            del r
