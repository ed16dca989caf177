"""Deterministic synthetic inputs: fully developed speckle with shot noise.

Host (NumPy) generators are used by the parity tests, the golden-vector script and the
CPU baseline; `speckle_stack_device` builds large stacks directly in HBM with torch so
that bench inputs are resident before the timed region starts.
"""
from __future__ import annotations

import numpy as np


def speckle_intensity(n: int, seed: int, *, pupil_div: int = 8, mean: float = 1000.0) -> np.ndarray:
    """Noise-free speckle intensity (n, n) float64: |ifft2(pupil * exp(2 pi i U))|^2, mean `mean`."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[-(n // 2):n - n // 2, -(n // 2):n - n // 2]
    pupil = (xx * xx + yy * yy) <= (n / pupil_div) ** 2
    field = np.fft.ifft2(np.fft.ifftshift(pupil * np.exp(2j * np.pi * rng.random((n, n)))))
    inten = np.abs(field) ** 2
    return inten * (mean / inten.mean())


def speckle_frame(n: int, seed: int, *, pupil_div: int = 8, mean: float = 1000.0,
                  dtype=np.float32) -> np.ndarray:
    """One Poisson-noised speckle frame; frame t of a stack uses seed = 1234 + t (§8d)."""
    rng = np.random.default_rng(seed + 7_000_000)
    return rng.poisson(speckle_intensity(n, seed, pupil_div=pupil_div, mean=mean)).astype(dtype)


def speckle_stack(t: int, n: int, *, seed0: int = 1234, dtype=np.float32) -> np.ndarray:
    return np.stack([speckle_frame(n, seed0 + i, dtype=dtype) for i in range(t)], axis=0)


def spiral_shifts(t: int, max_shift: int = 32) -> np.ndarray:
    """Integer (sy, sx) spiral, |s| <= max_shift, shifts[0] = (0, 0)."""
    k = np.arange(t)
    r = max_shift * k / max(t - 1, 1)
    ang = 0.61803398875 * 2 * np.pi * k
    s = np.stack([np.rint(r * np.sin(ang)), np.rint(r * np.cos(ang))], axis=1).astype(np.int64)
    s[0] = 0
    return s


def shifted_stack(t: int, n: int, *, seed: int = 1234, max_shift: int = 32, dtype=np.float32):
    """cfg3 stack: frame k = Poisson(roll(I0, shift_k)), fresh shot noise per frame.

    Returns (stack (t, n, n), shifts (t, 2) int64 with ground-truth (sy, sx))."""
    i0 = speckle_intensity(n, seed)
    sh = spiral_shifts(t, max_shift)
    out = np.empty((t, n, n), dtype=dtype)
    for k in range(t):
        rng = np.random.default_rng(5000 + k if k else seed)
        out[k] = rng.poisson(np.roll(i0, (int(sh[k, 0]), int(sh[k, 1])), axis=(0, 1))).astype(dtype)
    return out, sh


def speckle_stack_device(t: int, n: int, *, seed0: int = 1234, device="cuda", chunk: int = 8,
                         pupil_div: int = 8, mean: float = 1000.0):
    """(t, n, n) float32 speckle stack generated in HBM (same statistics as `speckle_frame`,
    different random stream: torch's counter-based generator).  Used for bench inputs only;
    parity tests use the host generators.  The generator itself calls torch.fft (rocFFT shows up as
    `fft_rtc_*` kernels in profiles of the input set-up): it builds test data outside every timed region and is
    not part of the product path, which has no rocFFT dependency."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed0)
    ax = torch.arange(n, device=device) - n // 2
    pupil = ((ax[None, :] ** 2 + ax[:, None] ** 2) <= (n / pupil_div) ** 2).to(torch.float32)
    pupil = torch.fft.ifftshift(pupil)
    out = torch.empty((t, n, n), dtype=torch.float32, device=device)
    for a in range(0, t, chunk):
        b = min(t, a + chunk)
        ph = torch.rand((b - a, n, n), generator=g, device=device) * (2 * np.pi)
        fld = torch.fft.ifft2(torch.polar(pupil.expand(b - a, n, n), ph))
        inten = fld.real ** 2 + fld.imag ** 2
        inten *= mean / inten.mean(dim=(1, 2), keepdim=True)
        out[a:b] = torch.poisson(inten, generator=g)
    return out
