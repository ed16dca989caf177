#!/usr/bin/env python3
"""bench.py -- headline benchmark of the barc4dip hot path on MI355X.

Workload (BASELINE.json configs[1]): 2-D FFT -> PSD -> autocorrelation on a 256-frame
2048x2048 float32 synthetic speckle stack per GPU, resident in HBM before timing.
One "step" = one pass of b4d_psd_autocorr2d over the whole stack (PSD + autocorrelation
written for every frame).  Metric: frames/s (whole job, all ranks), plus
  * roofline: the dominant kernel (column FFT/PSD/inverse) priced by HIP events inside the
    timed region, against the 8 TB/s HBM roof (DESIGN.md states the byte accounting);
  * cpu_baseline: the NumPy oracle (port of the reference's fft2d + psd2d + autocorr2d calls)
    timed on the host cores of this box on a bounded sample.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N = 2048
FRAMES_PER_GPU = 256
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def baseline_metric() -> str:
    """BASELINE.json's own metric string (frames/s is `value`; the roofline share is in `roofline` / `pipeline_roofline`)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "frames/s (2048\u00d72048 fp32) through FFT\u2192PSD\u2192autocorr; % HBM roofline"


def algorithmic_bytes(n: int):
    """Per-frame byte model.  pipe = SURVEY.md §8(d) 4-pass figure (12 N^2 + 40 N Nh);
    col = what the fused column kernel must move: half spectrum in (8 N N/2) + PSD out (4 N^2) + rows 0..N/2+1 of the
    inverse-column output (8 (N/2+2) N/2; the autocorrelation is even, the other rows are never read)."""
    nh = n // 2 + 1
    return {"pipe": 12 * n * n + 40 * n * nh, "col": 4 * n * n + 4 * n * n + 8 * (n // 2 + 2) * (n // 2),
            "r2c": 8 * n * n, "c2r": 4 * (n // 2 + 2) * n + 4 * n * n}


def cpu_baseline(n: int, frames: int):
    """The reference's three public calls per frame (fft2d, psd2d, autocorr2d) via the NumPy oracle."""
    from barc4dip_amd import synth
    from oracle import signal_np as S

    imgs = [synth.speckle_frame(n, 1234 + i) for i in range(frames)]
    t0 = time.perf_counter()
    for im in imgs:
        S.fft2d(im)
        S.psd2d(im)
        S.autocorr2d(im)
    dt = time.perf_counter() - t0
    out = {"value": frames / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{frames} frames of {n}x{n} float32: oracle fft2d + psd2d + autocorr2d per frame, 1 process, "
                     f"NumPy {np.__version__} pocketfft single-threaded"}
    try:
        from joblib import Parallel, delayed

        nthreads = min(len(os.sched_getaffinity(0)), 16)
        many = imgs * max(1, (2 * nthreads) // max(1, frames))

        def one(im):
            S.fft2d(im)
            S.psd2d(im)
            S.autocorr2d(im)

        t0 = time.perf_counter()
        Parallel(n_jobs=nthreads, prefer="threads")(delayed(one)(im) for im in many)
        dt = time.perf_counter() - t0
        out["threads"] = {"value": len(many) / dt, "unit": "frames/s", "cores": nthreads,
                          "note": "joblib threads over frames, as the reference's stack functions do"}
    except Exception as e:  # pragma: no cover
        out["threads"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU (default: the cfg2 stack)")
    ap.add_argument("--chunk", type=int, default=0, help="frames per launch group (0 = library default)")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))   # modulo: lets a 1-GPU box rehearse N > 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one node by contract: keep every rendezvous on the loopback interface (the container hostname may not resolve)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        # one process per GPU; RCCL ("nccl") carries device collectives (none on this data path: frames are
        # sharded, nothing is exchanged), gloo carries the host-side barrier and the max-over-ranks of the timing
        dist.init_process_group(backend="cpu:gloo,cuda:nccl")

    from barc4dip_amd import _ffi, synth

    T = args.frames
    stack = synth.speckle_stack_device(T, N, seed0=1234 + 100000 * rank)
    psd = torch.empty_like(stack)
    ac = torch.empty_like(stack)
    chunk = args.chunk or _ffi.default_chunk(N, N)
    plan = _ffi.Plan(N, N, chunk)
    lib = _ffi.lib()
    kms = (C.c_float * 4)()
    flags = _ffi.REMOVE_MEAN | _ffi.NORM_PEAK
    call = (plan.handle, C.c_void_p(stack.data_ptr()), T, C.c_void_p(psd.data_ptr()), 1.0 / (N * N),
            C.c_void_p(ac.data_ptr()), flags, _ffi.stream_ptr())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:   # host-side rendezvous (gloo): this data path has no device collective to piggy-back on
            dist.all_reduce(torch.zeros(1, dtype=torch.float64))
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        _ffi.check(lib.b4d_psd_autocorr2d(*call))
    barrier()
    for i in range(4):
        kms[i] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _ffi.check(lib.b4d_psd_autocorr2d_timed(*call, kms))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # sanity on the product of the timed region (not a parity test): peak == 1 at the centre of every frame
    centre = ac[:, N // 2, N // 2]
    ok = bool(torch.all(centre == 1.0).item()) and bool(torch.isfinite(psd[0]).all().item())

    if rank == 0:
        B = algorithmic_bytes(N)
        frames_total = T * world * args.steps
        fps = frames_total / dt
        nlaunch = args.steps * ((T + chunk - 1) // chunk)            # launches of each kernel on this rank
        col_ms = kms[1] / nlaunch
        frames_per_launch = T / ((T + chunk - 1) // chunk)
        col_gbs = B["col"] * frames_per_launch / (col_ms * 1e-3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_col.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": baseline_metric(),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg2: 2D FFT->PSD->autocorr, {T}-frame {N}x{N} fp32 stack per GPU, "
                                   "PSD + autocorr written per frame",
                       "frames_per_gpu": T, "chunk": chunk, "parallelism": f"frames sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "kernel": "k_col (column FFT + |F|^2 PSD + inverse column FFT, fused)",
                         "achieved": col_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": col_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "bytes_per_launch": B["col"] * frames_per_launch,
                         "avg_launch_ms": col_ms,
                         "kernel_ms_per_step": {"row_r2c": kms[0] / args.steps, "col": kms[1] / args.steps,
                                                "peak": kms[2] / args.steps, "row_c2r": kms[3] / args.steps}},
            "pipeline_roofline": {"bytes_per_frame": B["pipe"], "achieved": B["pipe"] * fps / world / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s per GPU",
                                  "frac": B["pipe"] * fps / world / 1e9 / HBM_PEAK_GBS,
                                  "note": "SURVEY.md §8(d): B_pipe = 12N^2 + 40 N (N/2+1) per frame"},
            "outputs_ok": ok,
        }
        if not args.no_cpu and world == 1:     # the CPU leg is timed at N = 1 only (it would idle the other ranks)
            line["cpu_baseline"] = cpu_baseline(N, args.cpu_frames)
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        dist.all_reduce(torch.zeros(1, dtype=torch.float64))
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
