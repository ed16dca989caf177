"""Wall-clock of the aggregators (speckle_stats / sharpness_stats, tiles on) on the GPU path vs the oracle (CPU)."""
import json
import sys
import time
import warnings

import numpy as np
import torch

sys.path.insert(0, ".")
from barc4dip_amd import metrics as gm, synth  # noqa: E402
from oracle import metrics_np as M  # noqa: E402

warnings.simplefilter("ignore")
for n in (512, 2048):
    img = synth.speckle_frame(n, 1234)
    res = {}
    for name, gfn, cfn in (("speckle_stats", gm.speckle_stats, M.speckle_stats), ("sharpness_stats", gm.sharpness_stats, M.sharpness_stats)):
        gfn(img, verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gfn(img, verbose=False)
        torch.cuda.synchronize()
        tg = time.perf_counter() - t0
        t0 = time.perf_counter()
        ref = cfn(img)
        tc = time.perf_counter() - t0
        res[name] = {"gpu_s": round(tg, 4), "cpu_oracle_s": round(tc, 3), "tile_mode": out["meta"].get("tile_mode"),
                     "tiles_groups": sorted(out.get("tiles", {}))}
    print(json.dumps({"n": n, **res}), flush=True)
