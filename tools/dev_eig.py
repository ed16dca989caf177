"""dev: time + check b4d_sta2_eigenvalues."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from barc4dip_amd import synth
from barc4dip_amd.metrics import sharpness as SH
for (b, n) in ((1, 2048), (81, 228), (9, 683), (1, 512)):
    fr = np.stack([synth.speckle_frame(n, 40 + i) for i in range(b)]).astype(np.float32)
    d = torch.from_numpy(fr).cuda()
    SH._sta2_device(d); torch.cuda.synchronize()
    t0 = time.perf_counter(); e = SH._sta2_device(d); torch.cuda.synchronize(); t1 = time.perf_counter()
    x = d[:min(b, 4)].double(); j = x / torch.sqrt((x * x).sum(dim=(1, 2), keepdim=True)); j = j - j.mean(dim=(1, 2), keepdim=True)
    t2 = time.perf_counter(); sv = torch.linalg.svdvals(j); r = ((sv * sv) / float(j[0].numel() - 1)).cpu().numpy(); t3 = time.perf_counter()   # dense SVD, comparison only
    err = np.abs(e[:min(b, 4)] - r[:, :8]) / r[:, :8]
    print(f"b={b} n={n}: sta2 {1e3*(t1-t0):.2f} ms; svd({min(b,4)}) {1e3*(t3-t2):.1f} ms; max rel err {err.max():.2e}", flush=True)
