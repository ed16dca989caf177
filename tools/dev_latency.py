"""dev: single-call latency of the per-frame entry points (what an interactive user or a joblib worker sees)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import signal as gs, synth  # noqa: E402
from barc4dip_amd.preprocessing import deconvolve_psf  # noqa: E402

for n in (512, 1024, 2048):
    a = torch.from_numpy(synth.speckle_frame(n, 1)).cuda()
    b = torch.roll(a, (3, -2), (0, 1))
    sl = (slice(n // 2 - 60, n // 2 + 61), slice(n // 2 - 60, n // 2 + 61))
    tpl = a[sl].contiguous()
    calls = {
        "psd2d": lambda: gs.psd2d(a, return_tensors=True),
        "autocorr2d": lambda: gs.autocorr2d(a, return_tensors=True),
        "xcorr2d": lambda: gs.xcorr2d(a, b, return_tensors=True),
        "phase_correlation": lambda: gs.phase_correlation(tpl, b, slices_yx=sl),
        "deconvolve_psf": lambda: deconvolve_psf(a, sigma=1.5, return_tensors=True),
    }
    row = []
    for name, fn in calls.items():
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        row.append(f"{name} {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
    print(f"{n}^2: " + " | ".join(row), flush=True)
