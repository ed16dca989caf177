import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from barc4dip_amd import synth
from barc4dip_amd import _ffi
if len(sys.argv) > 1:
    _ffi._lib = _ffi.load_library(sys.argv[1])
from barc4dip_amd.metrics import kernels as K
for n, b in ((2048, 1), (1024, 4), (228, 81)):
    st = torch.from_numpy(np.stack([synth.speckle_frame(n, 5 + i) for i in range(min(b, 4))])).cuda()
    st = st.repeat((b + 3) // 4, 1, 1)[:b].contiguous()
    K.percentiles_batch(st, [0.05, 99.95]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r = K.percentiles_batch(st, [0.05, 99.95])
    torch.cuda.synchronize()
    ref = np.nanpercentile(st[0].cpu().numpy().astype(np.float64), [0.05, 99.95])
    print(f"percentiles {b}x{n}^2: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms  exact {np.array_equal(r[0], ref)}", flush=True)
