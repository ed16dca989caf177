import sys, time, warnings
sys.path.insert(0, ".")
import numpy as np, torch
from barc4dip_amd import metrics as gm, synth
warnings.simplefilter("ignore")
n, T = 2048, 16
stack = np.stack([synth.speckle_frame(n, 60 + i) for i in range(4)] * (T // 4))
kw = dict(tracking_method="phase", tracking_backend="internal", roi_grain_factor=20.0)
for m in ("all", ("amplitude", "stats", "bandwidth"), ("grain",), ("stats",)):
    gm.speckle_stack_stats(stack[:4], metrics=m, verbose=False, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); gm.speckle_stack_stats(stack, metrics=m, verbose=False, **kw); torch.cuda.synchronize()
    print(m, f"{(time.perf_counter()-t0)/T*1e3:.2f} ms/frame", flush=True)
