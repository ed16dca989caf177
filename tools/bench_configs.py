"""Secondary measurements for BASELINE.json configs 3 and 4 on one GPU (not the bench.py headline).

cfg3: phase-correlation tracking, 3x3 ROI grid abs + inc (18 pairs per frame) on a 1024x1024 stack.
cfg4: temporal per-pixel statistics of a 2048x2048 stack (single-GPU shard; the all-reduce is a no-op at N = 1).
Prints one JSON line per config, with the oracle timed on a bounded sample as CPU baseline."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import synth  # noqa: E402
from barc4dip_amd.geometry import roi_grid_3x3  # noqa: E402
from barc4dip_amd.metrics import temporal_stats  # noqa: E402
from barc4dip_amd.signal import phase_correlation_batch  # noqa: E402


def cfg3(T=64, n=1024, side=121, steps=3):
    stack, sh = synth.shifted_stack(T, n, seed=1234, max_shift=32)
    dev = torch.from_numpy(stack).cuda()
    grid, _ = roi_grid_3x3((n, n), (side, side), (side // 2, side // 2))
    rois = [(s[0].start, s[0].stop, s[1].start, s[1].stop) for s in grid.ravel()]
    tpl_frame = [0] * 9 + [max(t - 1, 0) for t in range(T) for _ in range(9)]
    tpl_roi = rois + rois * T
    pair_img = [t for t in range(T) for _ in range(9)] * 2
    pair_tpl = [k for _ in range(T) for k in range(9)] + [9 + 9 * t + k for t in range(T) for k in range(9)]
    res = phase_correlation_batch(dev, dev, tpl_frame, tpl_roi, pair_img, pair_tpl)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(steps):
        t0 = time.perf_counter()
        res = phase_correlation_batch(dev, dev, tpl_frame, tpl_roi, pair_img, pair_tpl)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = bool(np.all(np.median(np.rint(res[:9 * T, 0]).reshape(T, 9), axis=1) == sh[:, 0]))
    from oracle import signal_np as S

    t0 = time.perf_counter()
    npairs_cpu = 0
    for t in range(2):
        for r in rois:
            sl = (slice(r[0], r[1]), slice(r[2], r[3]))
            S.phase_correlation(stack[0][sl], stack[t], slices_yx=sl)
            npairs_cpu += 1
    cpu = npairs_cpu / (time.perf_counter() - t0)
    print(json.dumps({"config": "cfg3 phase-correlation tracking", "frames": T, "n": n, "pairs": len(pair_img),
                      "pairs_per_s": len(pair_img) / best, "frames_per_s": T / best, "truth_recovered": ok,
                      "cpu_port_pairs_per_s": cpu, "cpu_sample": f"{npairs_cpu} pairs, 1 core"}), flush=True)


def cfg4(T=256, n=2048, steps=5):  # 256-frame shard fits the default test box quickly; the kernel rate does not depend on T
    dev = synth.speckle_stack_device(T, n)
    temporal_stats(dev, return_tensors=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(steps):
        t0 = time.perf_counter()
        temporal_stats(dev, return_tensors=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    from oracle import temporal_np as Tn

    host = dev[:32].cpu().numpy()
    t0 = time.perf_counter()
    Tn.temporal_stats(host)
    cpu = 32 / (time.perf_counter() - t0)
    print(json.dumps({"config": "cfg4 temporal mean/var/contrast (1 GPU shard)", "frames": T, "n": n,
                      "frames_per_s": T / best, "GBps": T * n * n * 4 / best / 1e9, "cpu_port_frames_per_s": cpu,
                      "cpu_sample": "32 frames, NumPy float64 mean/var"}), flush=True)


def cfg5(T=32, n=4096, sigma=1.5, steps=3):   # T = 32 on one GPU
    from barc4dip_amd.preprocessing import deconvolve_psf

    dev = synth.speckle_stack_device(T, n)
    deconvolve_psf(dev[:1], sigma=sigma, return_tensors=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(steps):
        t0 = time.perf_counter()
        out = deconvolve_psf(dev, sigma=sigma, return_tensors=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    from oracle import wiener_np as W

    host = dev[0].cpu().numpy()
    t0 = time.perf_counter()
    ref = W.deconvolve_psf(host, sigma=sigma)
    cpu = 1 / (time.perf_counter() - t0)
    err = float(np.max(np.abs(out[0].cpu().numpy() - ref)) / np.max(np.abs(host)))
    print(json.dumps({"config": "cfg5 Wiener deconvolution (plan cached)", "frames": T, "n": n, "padded": n + 8,
                      "frames_per_s": T / best, "cpu_port_frames_per_s": cpu, "cpu_sample": "1 frame, NumPy float32 rfft2",
                      "max_err_vs_oracle_rel": err}), flush=True)


if __name__ == "__main__":
    import os
    if os.environ.get("B4D_NOPRED"):   # dev: time the fallback route of the tracker's median (no expected-bin gathering)
        from barc4dip_amd import _ffi
        _ffi.lib().b4d_set_option(b"track_predict_bin", 0)
    which = sys.argv[1:] or ["3", "4", "5"]
    if "3" in which:
        cfg3()
    if "4" in which:
        cfg4()
    if "5" in which:
        cfg5()
