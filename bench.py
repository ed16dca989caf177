#!/usr/bin/env python3
"""bench.py -- headline benchmark of the barc4dip hot path on MI355X.

Workload (BASELINE.json configs[1]): 2-D FFT -> PSD -> autocorrelation on a 256-frame
2048x2048 float32 synthetic speckle stack per GPU, resident in HBM before timing.
One "step" = one pass of b4d_psd_autocorr2d over the whole stack (PSD + autocorrelation
written for every frame).  Metric: frames/s (whole job, all ranks), plus
  * roofline: the dominant kernel (column FFT/PSD/inverse) priced by HIP events inside the
    timed region, against the 8 TB/s HBM roof (DESIGN.md states the byte accounting);
  * cpu_baseline: the NumPy oracle (port of the reference's fft2d + psd2d + autocorr2d calls)
    timed on the host cores of this box on a bounded sample;
  * secondary: the other BASELINE.json configs, measured AFTER the timed region of the headline
    (they never touch `value`): cfg3 phase-correlation tracking (1024^2), cfg4 temporal statistics
    (this rank's 2048^2 shard + ONE all-reduce over RCCL when --gpus N > 1) and cfg5 Wiener
    deconvolution (4096^2, sigma 1.5), each with its SURVEY.md §8(d) byte model, the fraction of the
    8 TB/s roof and the error against the oracle on a small sample.

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

# dmabuf IPC (the only mode the host driver of this pool supports): RCCL / cross-process tensor sharing fails with
# "hipIpcGetMemHandle: invalid argument" without it.  The environment normally exports it already; this is a seat belt.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N = 2048
FRAMES_PER_GPU = 256
BENCH_CHUNK = 256              # frames per launch group of the plan the bench creates (see main())
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_FILE = "profiles/r03_pmc_col.json"


def baseline_metric() -> str:
    """BASELINE.json's own metric string (frames/s is `value`; the roofline share is in `roofline` / `pipeline_roofline`)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "frames/s (2048×2048 fp32) through FFT→PSD→autocorr; % HBM roofline"


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


HBM_COPY_GBS = 6290.0          # MI355X_MICROARCH.md: what a float4 copy kernel reaches (the practical ceiling of a stream)


def fractions(model_bytes: float, moved_bytes: float, units_per_s: float) -> dict:
    """Three readings of one rate.  `model` prices SURVEY.md §8(d)'s pass-by-pass byte model (what BASELINE.md quotes; fusion
    can push it past the copy ceiling, so it is NOT a distance to the roof); `moved` prices the bytes the kernels of THIS
    implementation have to move (DESIGN.md's accounting, cross-checked by the FETCH_SIZE / WRITE_SIZE counters under profiles/);
    `frac_of_copy` is `moved` against the 6.29 TB/s a plain copy reaches on this part."""
    gm, gv = model_bytes * units_per_s / 1e9, moved_bytes * units_per_s / 1e9
    return {"model_bytes": int(model_bytes), "moved_bytes": int(moved_bytes), "model_GBps": gm, "moved_GBps": gv,
            "frac_model": gm / HBM_PEAK_GBS, "frac_moved": gv / HBM_PEAK_GBS, "frac_of_copy": gv / HBM_COPY_GBS}


# ---------------------------------------------------------------------------------------------- secondary configs
def _best_of(fn, sync, reps):
    fn()
    sync()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        sync()
        best = min(best, time.perf_counter() - t0)
    return best


def secondary_fft2d(torch, stack, cpu: bool):
    """SURVEY.md §8 row a1 (signal/fft.py:198-237): fftshift(fft2(frame)) of every frame of a resident stack -> complex64 full
    spectra, through the public fft2d_stack.  Model: 4 B in + 8 B out per pixel ("if fft2d's complex output is also
    materialised add 8 N^2").  Moved: the two passes of the implementation (b4d_spectrum.hip: real columns -> half spectrum along y,
    4 + 4 B per pixel; rows of it -> every output row and its conjugate mirror, 4 + 8 B per pixel).  The half spectrum between the
    passes (8 of those 20 B) is written and read inside ~64-MiB launch groups on two streams, i.e. through the memory-side cache
    (b4d_fft2d: Lanes); `moved` still counts it: it is what the kernels load and store."""
    from barc4dip_amd.signal.fft import fft2d_stack

    T, n = min(int(stack.shape[0]), 256), int(stack.shape[-1])
    sub = stack[:T]
    res = {}

    def run():
        res["out"] = fft2d_stack(sub, return_tensors=True)

    best = _best_of(run, torch.cuda.synchronize, 3)
    model = 12 * n * n
    moved = (4 + 4 + 4 + 8) * n * n
    line = {"workload": f"fft2d (row a1): {T} resident frames of {n}x{n} float32 -> shifted complex64 spectra",
            "frames_per_s": T / best, **fractions(model, moved, T / best)}
    if cpu:
        from oracle import signal_np as S

        t0 = time.perf_counter()
        ref = S.fft2d(sub[0].cpu().numpy().astype(np.float64))[0]
        dt = time.perf_counter() - t0
        got = res["out"][0].cpu().numpy()
        line.update({"max_err_vs_oracle_over_peak": float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))),
                     "cpu_port_frames_per_s": 1.0 / dt, "cpu_sample": "1 frame, float64 oracle, 1 core"})
    del res
    torch.cuda.empty_cache()
    return line


def secondary_cfg3(torch, cpu: bool):
    """BASELINE configs[2]: phase-correlation tracking on a 1024 x 1024 stack, SURVEY.md §8(d) protocol: 3 x 3 ROI grid,
    "abs" (template from frame 0) and "inc" (template from the previous frame) = 18 pairs per frame.  64 frames of the
    1024-frame stack are resident (the rate does not depend on T: every frame is transformed once, every pair once)."""
    from barc4dip_amd import synth
    from barc4dip_amd.geometry import roi_grid_3x3
    from barc4dip_amd.signal import phase_correlation_batch

    T, n, side = 64, 1024, 121
    stack, sh = synth.shifted_stack(T, n, seed=1234, max_shift=32)
    dev = torch.from_numpy(stack).cuda()
    grid, _ = roi_grid_3x3((n, n), (side, side), (side // 2, side // 2))
    rois = [(s[0].start, s[0].stop, s[1].start, s[1].stop) for s in grid.ravel()]
    tpl_frame = [0] * 9 + [max(t - 1, 0) for t in range(T) for _ in range(9)]
    tpl_roi = rois + rois * T
    pair_img = [t for t in range(T) for _ in range(9)] * 2
    pair_tpl = [k for _ in range(T) for k in range(9)] + [9 + 9 * t + k for t in range(T) for k in range(9)]
    res = {}

    def run():
        res["out"] = phase_correlation_batch(dev, dev, tpl_frame, tpl_roi, pair_img, pair_tpl)

    best = _best_of(run, torch.cuda.synchronize, 3)
    out = res["out"]
    npairs = len(pair_img)
    truth = bool(np.all(np.median(np.rint(out[:9 * T, 0]).reshape(T, 9), axis=1) == sh[:, 0]) and
                 np.all(np.median(np.rint(out[:9 * T, 1]).reshape(T, 9), axis=1) == sh[:, 1]))
    nh, hh = n // 2 + 1, n // 2
    model = 12 * n * n + 56 * n * nh
    # bytes this implementation moves per pair (profiles/*pmc_cfg3*): every image spectrum is built once and shared
    # by its 18 pairs (row pass 4n^2 + 4n^2, column pass in place 4n^2 + 4n^2); a template is row-transformed once per (frame, ROI)
    # from its `side` non-zero rows (side^2*4 in, side*n/2*8 out) and serves the "abs" pairs of all frames or one "inc" pair; per
    # pair the cross-power column pass reads the image's half spectrum and the template's `side` rows (its column transform
    # happens in registers), writes one half spectrum, and the inverse row pass reads it and writes no map (arg-max, median bin
    # and 3 rows only).
    ntpl = len(tpl_frame)
    moved = (T * 16 * n * n + ntpl * (4 * side * side + 8 * side * hh)) / npairs + 8 * n * hh + 8 * side * hh + 2 * 8 * n * hh
    line = {"workload": f"cfg3: phase-correlation tracking, {T} resident frames of {n}x{n}, 3x3 ROI grid abs + inc ({npairs} pairs)",
            "pairs_per_s": npairs / best, "frames_per_s": T / best, **fractions(model, moved, npairs / best),
            "ground_truth_recovered": truth}
    if cpu:
        from oracle import signal_np as S

        t0 = time.perf_counter()
        worst_sub, int_ok, k = 0.0, True, 0
        for t in (1, T - 1):
            for ri in (0, 4):
                r = rois[ri]
                sl = (slice(r[0], r[1]), slice(r[2], r[3]))
                ref = S.phase_correlation(stack[0][sl].astype(np.float64), stack[t].astype(np.float64), slices_yx=sl)
                got = out[9 * t + ri]
                int_ok &= (round(ref[0]) == round(got[0])) and (round(ref[1]) == round(got[1]))
                worst_sub = max(worst_sub, abs(ref[0] - got[0]), abs(ref[1] - got[1]))
                k += 1
        dt = time.perf_counter() - t0
        line.update({"max_subpixel_err_vs_oracle_px": worst_sub, "integer_shifts_equal_oracle": bool(int_ok),
                     "cpu_port_pairs_per_s": k / dt, "cpu_sample": f"{k} pairs, float64 oracle, 1 core"})
    return line


def secondary_cfg5(torch, cpu: bool):
    """BASELINE configs[4]: Gaussian-PSF Wiener deconvolution of 4096 x 4096 frames, sigma 1.5 (padded 4104^2), T = 32
    resident frames, through the public deconvolve_psf with device tensors in and out."""
    from barc4dip_amd import synth
    from barc4dip_amd.preprocessing import deconvolve_psf

    T, n, sigma = 32, 4096, 1.5
    dev = synth.speckle_stack_device(T, n, seed0=4321)
    res = {}

    def run():
        res["out"] = deconvolve_psf(dev, sigma=sigma, return_tensors=True)

    best = _best_of(run, torch.cuda.synchronize, 3)
    m, mh = n + 8, (n + 8) // 2 + 1
    model = 12 * m * m + 48 * m * mh + 4 * n * n
    # moved (counters under profiles/): rows forward 4 n^2 in + 8 m mh out, columns 8 m mh in + 8 m mh (complex
    # filter) + 8 m mh out, rows inverse 8 m mh in + 4 n^2 out = 67 + 67, 67 + 67 + 67, 67 + 67 MB at 4096^2 (no padded copy,
    # no transposes, no max pass)
    moved = 4 * n * n + 8 * m * mh + 3 * 8 * m * mh + 8 * m * mh + 4 * n * n
    line = {"workload": f"cfg5: Wiener deconvolution, {T} resident frames of {n}x{n}, sigma {sigma} (padded {m}x{m} = 8*27*19)",
            "frames_per_s": T / best, **fractions(model, moved, T / best)}
    if cpu:
        from oracle import wiener_np as W

        host = dev[0].cpu().numpy()
        t0 = time.perf_counter()
        ref = W.deconvolve_psf(host, sigma=sigma)
        dt = time.perf_counter() - t0
        line.update({"max_err_vs_oracle_over_range": float(np.max(np.abs(res["out"][0].cpu().numpy() - ref)) / np.max(np.abs(host))),
                     "cpu_port_frames_per_s": 1.0 / dt, "cpu_sample": "1 frame, float32 oracle (parity unpinned: scikit-image absent), 1 core"})
    del dev, res
    torch.cuda.empty_cache()
    return line


def secondary_cfg4(torch, dist, stack, world: int, rank: int, cpu: bool):
    """BASELINE configs[3]: per-pixel temporal mean / variance / contrast of 8192 x 2048 x 2048 frames sharded over 8 GPUs =
    1024 frames (16 GiB) per GPU.  Every rank streams ITS 1024 frames into float64 sums, then ONE all-reduce of
    [count, sum_x, sum_xx] (64 MiB) crosses xGMI (RCCL) and the maps are finalised from the reduced sums.  Reported:
    whole-job frames/s, the collective's own time, and equality of the N-rank result with a single-rank run on a small stack."""
    from barc4dip_amd.metrics.temporal import shard_bounds, temporal_stats

    T, H, W = (int(v) for v in stack.shape)
    tm = {}

    def run():
        temporal_stats(stack, return_tensors=True, timings=tm if world > 1 else None)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.all_reduce(torch.zeros(1, dtype=torch.float64))   # gloo: host-side rendezvous

    run()
    sync()
    best = 1e30
    for _ in range(3):
        sync()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        best = min(best, dt)
    bytes_frame = 4 * H * W
    alt = None
    if world > 1:      # SURVEY.md section 8e's all-links alternative beside the north-star's single all-reduce: reduce-scatter of row
        ts = {}        # slices, local finalisation, all-gather of the float32 maps (same maps, bit for bit)
        best_s = 1e30
        for _ in range(3):
            sync()
            t0 = time.perf_counter()
            temporal_stats(stack, return_tensors=True, collective="reduce_scatter", timings=ts)
            torch.cuda.synchronize()
            dts = time.perf_counter() - t0
            tmax = torch.tensor([dts], dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            best_s = min(best_s, float(tmax.item()))
        alt = {"frames_per_s": T * world / best_s, "reduce_scatter_ms": ts.get("reduce_scatter_ms"), "all_gather_ms": ts.get("all_gather_ms"),
               "payload_bytes": {"reduce_scatter_in": 8 * world * (2 + 2 * (-(-H // world)) * W), "all_gather_out": 4 * 3 * world * (-(-H // world)) * W}}
    line = {"workload": f"cfg4: temporal mean/var/contrast, {T} frames of {H}x{W} per GPU x {world} GPU(s), "
                        + ("one RCCL all-reduce of 2*H*W+2 float64" if world > 1 else "no collective at N = 1"),
            "frames_per_s": T * world / best, "per_gpu": fractions(bytes_frame, bytes_frame, T / best),
            "allreduce_ms": tm.get("allreduce_ms"), "allreduce_payload_bytes": 8 * (2 * H * W + 2) if world > 1 else 0,
            "reduce_scatter_all_gather_route": alt}
    # correctness on a small stack that every rank can generate: N-rank sharded result == single-rank result, and both
    # against the float64 NumPy expressions (oracle/temporal_np.py)
    Ts = 8 * world + 3
    rng = np.random.default_rng(77)
    small = rng.poisson(900.0, size=(Ts, 96, 128)).astype(np.float32)
    t0, t1 = shard_bounds(Ts, world, rank)
    got = [x.cpu().numpy() for x in temporal_stats(torch.from_numpy(small[t0:t1]).cuda(), return_tensors=True, overlap_chunks=3)]
    if rank == 0:
        if world > 1:
            solo = dist.new_group([0])
        one = [x.cpu().numpy() for x in temporal_stats(torch.from_numpy(small).cuda(), return_tensors=True,
                                                        group=solo if world > 1 else None)]
        line["n_rank_equals_single_rank"] = bool(all(np.array_equal(a, b) for a, b in zip(got, one)))
        if cpu:
            from oracle import temporal_np as Tn

            ref = Tn.temporal_stats(small)
            line["max_rel_err_vs_oracle"] = float(max(np.max(np.abs(a - r) / np.maximum(np.abs(r), 1e-30)) for a, r in zip(got, ref)))
            host = stack[:16].cpu().numpy()
            tc = time.perf_counter()
            Tn.temporal_stats(host)
            line["cpu_port_frames_per_s"] = 16 / (time.perf_counter() - tc)
            line["cpu_sample"] = "16 frames of 2048x2048, NumPy float64 sums, 1 core"
    elif world > 1:
        dist.new_group([0])   # collective: every rank takes part in the creation of rank 0's solo group
    return line


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU (default: the cfg2 stack)")
    ap.add_argument("--chunk", type=int, default=0, help="frames per launch group (0 = library default)")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--tune", type=int, default=0, help="workspace candidates of b4d_plan_tune during warm-up (0 / 1 = off, the default: "
                                                        "the library places its workspaces itself)")
    ap.add_argument("--tune-compare", type=int, default=0, help="after the timed region (N = 1): candidates of a b4d_plan_tune "
                                                                "comparison run reported beside `value` (0 / 1 = off, the default: "
                                                                "profiles/r03_placement.txt -- the tuner buys nothing once the GPU is warm)")
    ap.add_argument("--preheat", type=float, default=0.6,
                    help="seconds of untimed passes BEFORE the W warm-up steps: the first process on an idle GPU runs the column and "
                         "inverse row kernels 6-7 %% slower for its first ~0.2 s (profiles/r03_placement.txt)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the fft2d / cfg3 / cfg4 / cfg5 block after the timed region")
    ap.add_argument("--spawn-check", action="store_true",
                    help="rendezvous only: every rank joins the host-side (gloo) group, rank 0 prints what it saw; no GPU call")
    return ap.parse_args(argv)


def free_port() -> int:
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: THIS process (which has not imported torch.cuda and never touches the
    GPU) starts N children of the same command line, one rank per GPU, with the torch.distributed environment of a one-node
    job on the loopback interface, relays rank 0's stdout (the JSON line) and returns non-zero if any rank did.  No exec, no
    restart of a process that has initialised the GPU."""
    import subprocess
    import tempfile

    port = free_port()
    procs = []
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), B4D_BENCH_SPAWNED="1")
            env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        codes = [None] * n
        while any(c is None for c in codes):      # a rank that dies takes the job down (the others would wait in a collective)
            for r, p in enumerate(procs):
                if codes[r] is None:
                    codes[r] = p.poll()
            if any(c not in (None, 0) for c in codes):
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        p.terminate()
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        try:
                            codes[r] = p.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            p.kill()
                            codes[r] = p.wait()
                break
            time.sleep(0.05)
        out0.seek(0)
        sys.stdout.write(out0.read().decode("utf-8", "replace"))
        sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv))      # nothing below runs in the parent: it never sees the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:      # a job of another size than the one asked for must fail, not degrade to what is there
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as n_gpus={args.gpus}")

    # ONE JSON line on stdout: libraries underneath (gloo's "[Gloo] Rank 0 is connected to ..." lines, RCCL banners) write to file
    # descriptor 1 from C++, so everything but the line itself is sent to stderr from here on
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj) -> None:
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist

    backend = os.environ.get("B4D_BENCH_BACKEND", "cpu:gloo,cuda:nccl")
    rehearsal = "nccl" not in backend        # B4D_BENCH_BACKEND=gloo: N ranks on fewer GPUs (RCCL refuses duplicate devices);
    if world > 1:                             # never set by the driver
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one node by contract: keep every rendezvous on the loopback interface (the container hostname may not resolve)
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    if args.spawn_check:
        if os.environ.get("B4D_BENCH_TEST_FAIL_RANK") == str(rank):      # tests/test_bench_spawn.py: a rank that dies before the rendezvous
            sys.exit(7)
        if world > 1:
            dist.init_process_group(backend="gloo")
        seen = torch.tensor([float(rank + 1)], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(seen)
        if rank == 0:
            emit({"spawn_check": True, "n_gpus": world, "dist_world_size": dist.get_world_size() if world > 1 else 1,
                  "rank_sum": float(seen.item()), "launcher": "bench.py" if os.environ.get("B4D_BENCH_SPAWNED") else "external"})
        if world > 1:
            dist.destroy_process_group()
        return
    ndev = torch.cuda.device_count()
    if ndev < world and not rehearsal:
        raise SystemExit(f"--gpus {world} but only {ndev} device(s) visible (B4D_BENCH_BACKEND=gloo rehearses N ranks on fewer GPUs)")
    torch.cuda.set_device(local_rank % max(1, ndev))
    if world > 1:
        # one process per GPU; RCCL ("nccl") carries device collectives (the cfg4 all-reduce; the cfg2 data path has none:
        # frames are sharded, nothing is exchanged), gloo carries the host-side barrier and the max-over-ranks of the timing
        dist.init_process_group(backend=backend)

    from barc4dip_amd import _ffi, synth

    T = args.frames
    stack = synth.speckle_stack_device(T, N, seed0=1234 + 100000 * rank)
    psd = torch.empty_like(stack)
    ac = torch.empty_like(stack)
    # frames per launch group: the whole per-GPU stack (plan workspace 16.8 MB per frame = 4.3 GB of the 288 GB).  The library
    # default for this size is 64 (1 GiB of workspace for callers who did not ask): measured 43.7-44.0 k frames/s against
    # 45.2-45.7 k with 128-256 frames per group -- the column pass walks 16 384 tiles per launch instead of 4 096.
    chunk = args.chunk or min(T, BENCH_CHUNK)
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

    # `value` is measured on the plan as the library hands it out (no bench-only tuning: --tune 0 is the default).  --tune K > 1
    # runs b4d_plan_tune BEFORE the warm-up instead (K workspace candidates, fastest kept: DESIGN.md §8.6) -- an experiment knob.
    tuned = None
    if args.tune > 1:
        tuned = plan.tune(stack, psd, ac, psd_scale=1.0 / (N * N), flags=flags, candidates=args.tune)
    # device spin-up: a GPU that has been idle (a fresh box: the driver's case) reaches its steady state only after a few hundred
    # milliseconds of load -- profiles/r03_placement.txt: first process on a box 45.7 k frames/s with W = 5, the same plan and
    # tensors 47.8 k 100 ms later; processes 2-5 47.8-47.9 k from the start.  Untimed, before the W warm-up steps, disclosed in `config`.
    preheat_passes = 0
    t_pre = time.perf_counter()
    while args.preheat > 0 and time.perf_counter() - t_pre < args.preheat:
        for _ in range(8):
            _ffi.check(lib.b4d_psd_autocorr2d(*call))
        torch.cuda.synchronize()
        preheat_passes += 8
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

    # after the timed region, N = 1 only: the same K steps once more on the workspace b4d_plan_tune keeps out of --tune-compare
    # candidates -- reported beside `value` so that the placement lottery is visible in the line, never as `value`
    tune_cmp = None
    if world == 1 and args.tune <= 1 and args.tune_compare > 1:
        try:
            tc = plan.tune(stack, psd, ac, psd_scale=1.0 / (N * N), flags=flags, candidates=args.tune_compare)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                _ffi.check(lib.b4d_psd_autocorr2d(*call))
            torch.cuda.synchronize()
            dtt = time.perf_counter() - t1
            tune_cmp = {"value_with_plan_tune": T * args.steps / dtt, "candidates": args.tune_compare, "kept_ms_per_pass": tc[0],
                        "slowest_ms_per_pass": tc[1], "note": "same K steps after b4d_plan_tune picked the fastest of the candidate "
                                                              "workspaces; measured after the timed region, not `value`"}
        except Exception as e:
            tune_cmp = {"error": repr(e)}
    plan.close()
    del psd, ac, plan
    torch.cuda.empty_cache()

    def all_ranks_ok(flag: bool) -> bool:
        """A collective phase runs only if EVERY rank is able to enter it (an OOM on one rank must not leave the others waiting)."""
        if world == 1:
            return flag
        t = torch.tensor([0.0 if flag else 1.0], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t.item()) == 0.0

    secondary = None
    frames_c4 = T
    if not args.no_secondary:
        cpu = not args.no_cpu
        secondary = {"note": "measured after the timed region of the headline; never part of `value`; HBM roof 8 TB/s, copy ceiling "
                             "6.29 TB/s; frac_model prices SURVEY.md §8(d)'s byte models, frac_moved the bytes this implementation moves"}
        if world == 1:
            try:
                secondary["fft2d"] = secondary_fft2d(torch, stack, cpu)
            except Exception as e:
                secondary["fft2d"] = {"error": repr(e)}
        # cfg4's per-GPU shard is 1024 frames: extend the 256-frame cfg2 stack (the generator is deterministic per seed)
        from barc4dip_amd import synth as _synth

        big = None
        try:
            big = torch.cat([stack, _synth.speckle_stack_device(1024 - T, N, seed0=777 + 100000 * rank)]) if T < 1024 else stack
        except Exception:      # not enough free HBM for 16 GiB on this rank
            big = None
        if all_ranks_ok(big is not None):      # every rank has the 1024-frame shard, or every rank keeps the 256-frame one
            stack = big
        del big
        frames_c4 = int(stack.shape[0])
        try:
            c4 = secondary_cfg4(torch, dist, stack, world, rank, cpu)
        except Exception as e:      # a rank-local failure is reported; the collective inside has its own agreement step
            c4 = {"error": repr(e)}
        if rank == 0:
            secondary["cfg4"] = c4
        if world == 1:
            del stack
            torch.cuda.empty_cache()
            for name, fn in (("cfg3", secondary_cfg3), ("cfg5", secondary_cfg5)):
                try:
                    secondary[name] = fn(torch, cpu)
                except Exception as e:      # a secondary leg never takes the headline line down with it
                    secondary[name] = {"error": repr(e)}
        else:
            secondary["fft2d"] = secondary["cfg3"] = secondary["cfg5"] = \
                "measured at --gpus 1 only (frames are sharded, no collective: N ranks run N copies)"

    if rank == 0:
        B = algorithmic_bytes(N)
        frames_total = T * world * args.steps
        fps = frames_total / dt
        nlaunch = args.steps * ((T + chunk - 1) // chunk)            # launches of each kernel on this rank
        col_ms = kms[1] / nlaunch
        frames_per_launch = T / ((T + chunk - 1) // chunk)
        col_gbs = B["col"] * frames_per_launch / (col_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        pmc_path = os.path.join(ROOT, PMC_FILE)
        if os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                pmc_frames = float(pmc.get("frames_per_launch", 64))
                traffic = pmc.get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic = traffic * frames_per_launch / pmc_frames      # counters are per launch: same launch size as `achieved`
                traffic_source = (f"{PMC_FILE} (static: a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run of this bench at "
                                  f"{pmc_frames:.0f} frames per launch, scaled to this run's {frames_per_launch:.0f}; NOT measured by the "
                                  "process that printed this line)")
            except Exception:
                traffic = None
        line = {
            "metric": baseline_metric(),
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg2: 2D FFT->PSD->autocorr, {T}-frame {N}x{N} fp32 stack per GPU, "
                                   "PSD + autocorr written per frame",
                       "frames_per_gpu": T, "chunk": chunk, "parallelism": f"frames sharded x{world}, no collective",
                       "preheat": {"seconds": args.preheat, "passes": preheat_passes,
                                   "note": "untimed passes before the W warm-up steps (idle-GPU spin-up; the timed region is exactly K steps)"},
                       "plan_tune": (None if tuned is None else
                                     {"candidates": args.tune, "kept_ms_per_pass": tuned[0], "slowest_ms_per_pass": tuned[1],
                                      "note": "b4d_plan_tune during warm-up (rank 0's figures): fastest of the plan's workspace "
                                              "allocations for this call, untimed"})},
            "roofline": {"bound": "hbm", "kernel": "k_col (column FFT + |F|^2 PSD + inverse column FFT, fused)",
                         "achieved": col_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": col_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": B["col"] * frames_per_launch,
                         "avg_launch_ms": col_ms,
                         "kernel_ms_per_step": {"row_r2c": kms[0] / args.steps, "col": kms[1] / args.steps,
                                                "peak": kms[2] / args.steps, "row_c2r": kms[3] / args.steps}},
            "pipeline_roofline": {**fractions(B["pipe"], B["r2c"] + B["col"] + B["c2r"], fps / world),
                                  "peak": HBM_PEAK_GBS, "copy_ceiling": HBM_COPY_GBS, "unit": "GB/s per GPU",
                                  "note": "model = SURVEY.md §8(d) B_pipe = 12N^2 + 40 N (N/2+1) per frame (BASELINE.md's figure; the "
                                          "north-star bar is frac_model >= 0.30); moved = what the three kernels of this implementation "
                                          "move (8 + 10 + 6) N^2: the fused column pass removed one round trip of the spectrum, so "
                                          "frac_model over-states the distance covered to the roof -- read frac_moved / frac_of_copy for that"},
            "distributed": {"world_size": dist.get_world_size() if world > 1 else 1,
                            "launcher": "bench.py (self-spawned ranks)" if os.environ.get("B4D_BENCH_SPAWNED") else
                                        ("torch.distributed.run / external" if world > 1 else "single process"),
                            "host_backend": (dist.get_backend() if world > 1 else None),
                            "device_collective_backend": (("nccl (RCCL)" if not rehearsal else "gloo (rehearsal: ranks share GPUs)")
                                                          if world > 1 else None),
                            "devices_visible": ndev,
                            "data_path_collectives": "cfg2: none (frames sharded); cfg4: ONE all-reduce of [count, sum x, sum x^2] "
                                                     "(secondary.cfg4.allreduce_ms)"},
            "outputs_ok": ok,
        }
        if tune_cmp is not None:
            line["plan_tune_comparison"] = tune_cmp
        if secondary is not None:
            line["secondary"] = secondary
        if not args.no_cpu and world == 1:     # the CPU leg is timed at N = 1 only (it would idle the other ranks)
            line["cpu_baseline"] = cpu_baseline(N, args.cpu_frames)
        emit(line)
    if world > 1:
        dist.all_reduce(torch.zeros(1, dtype=torch.float64))
        dist.destroy_process_group()
    if not ok:      # a wrong product of the timed region must not look like a measurement
        sys.exit(3)


if __name__ == "__main__":
    main()
