"""dev: soak of the two-lane launch groups: random sizes / stack lengths / entry points, option "lanes" 1 against 0, outputs must be
bit-identical; interleaved with work on the caller's stream before and after each call (stream-order check)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi, signal as gs  # noqa: E402
from barc4dip_amd.preprocessing import deconvolve_psf  # noqa: E402
from barc4dip_amd.signal.fft import fft2d_stack  # noqa: E402

lib = _ffi.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
pow2 = [256, 512, 1024, 2048]
wmr = [480, 512, 520, 576, 640, 720, 768, 1080, 1200, 1920]
t_end = time.time() + budget
n = bad = 0
while time.time() < t_end:
    kind = rng.choice(["fft_pow2", "fft_wmr", "pipe_wmr", "wiener", "wiener512"])
    if kind == "fft_pow2":
        ny, nx = int(rng.choice(pow2)), int(rng.choice(pow2))
    elif kind in ("fft_wmr", "pipe_wmr"):
        ny, nx = int(rng.choice(wmr)), int(rng.choice(wmr))
    elif kind == "wiener":
        ny = nx = 1024 + 8 * int(rng.integers(0, 2))       # 1024 -> 1032 (fused route), 1032 -> 1040 = 16 * 65 (mixed radix?)
    else:
        ny = nx = 512
    T = int(rng.integers(2, max(3, min(160, (768 << 20) // (ny * nx * 4)))))
    x = torch.rand((T, ny, nx), device="cuda") * 1000 + 1
    outs = []
    for lanes in (1, 0):
        assert lib.b4d_set_option(b"lanes", lanes) == 0
        y = x * 2.0                                         # queued on the caller's stream right before the call ...
        if kind.startswith("fft"):
            r = fft2d_stack(y, return_tensors=True)
        elif kind == "pipe_wmr":
            r = torch.stack(gs.psd_autocorr2d_stack(y, return_tensors=True)[:2])
        else:
            r = deconvolve_psf(y, sigma=1.5, return_tensors=True)
        r = r + 0                                           # ... and right after it
        y.zero_()                                           # would corrupt the input if the call's lanes were not joined
        outs.append(r.cpu())
    lib.b4d_set_option(b"lanes", 1)
    same = torch.equal(torch.view_as_real(outs[0]) if outs[0].is_complex() else outs[0],
                       torch.view_as_real(outs[1]) if outs[1].is_complex() else outs[1])
    n += 1
    if not same:
        bad += 1
        print("MISMATCH", kind, ny, nx, T, flush=True)
    del x, outs, r, y
    if n % 20 == 0:
        print(f"{n} cases, {bad} mismatches", flush=True)
print(f"done: {n} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
