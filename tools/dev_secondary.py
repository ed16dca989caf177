"""dev: the secondary legs of bench.py on their own (no headline): usage dev_secondary.py [fft2d] [cfg3] [cfg5] [pipe512] [pipe1024]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from barc4dip_amd import synth  # noqa: E402

what = sys.argv[1:] or ["fft2d", "cfg3", "cfg5"]
torch.cuda.set_device(0)
for w in what:
    if w == "fft2d":
        st = synth.speckle_stack_device(128, 2048, seed0=1234)
        r = bench.secondary_fft2d(torch, st, False)
        del st
    elif w == "cfg3":
        r = bench.secondary_cfg3(torch, False)
    elif w == "cfg5":
        r = bench.secondary_cfg5(torch, False)
    elif w.startswith("pipe"):
        from barc4dip_amd.signal import psd_autocorr2d_stack

        n = int(w[4:])
        T = max(16, (1 << 30) // (4 * n * n))
        st = synth.speckle_stack_device(T, n, seed0=1)
        psd_autocorr2d_stack(st, return_tensors=True)
        best = bench._best_of(lambda: psd_autocorr2d_stack(st, return_tensors=True), torch.cuda.synchronize, 5)
        r = {"workload": f"psd + autocorr {T} x {n}^2", "frames_per_s": T / best}
        del st
    torch.cuda.empty_cache()
    print(w, json.dumps({k: v for k, v in r.items() if k in ("frames_per_s", "pairs_per_s", "frac_model", "frac_moved", "ground_truth_recovered", "workload")}), flush=True)
