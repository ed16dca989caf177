"""dev: throughput of the NCC template tracker on the cfg3 protocol (64 x 1024^2, 3x3 ROI grid, abs + inc)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from barc4dip_amd import synth, _ffi as _f0
if len(sys.argv) > 1:
    _f0.LIB_PATH = sys.argv[1]
from barc4dip_amd.geometry import roi_grid_3x3
from barc4dip_amd.signal import phase_correlation_batch, template_matching_batch
T, n, side = 64, 1024, 121
stack, sh = synth.shifted_stack(T, n, seed=1234, max_shift=32)
dev = torch.from_numpy(stack).cuda()
grid, _ = roi_grid_3x3((n, n), (side, side), (side // 2, side // 2))
rois = [(s[0].start, s[0].stop, s[1].start, s[1].stop) for s in grid.ravel()]
tpl_frame = [0] * 9 + [max(t - 1, 0) for t in range(T) for _ in range(9)]
tpl_roi = rois + rois * T
pair_img = [t for t in range(T) for _ in range(9)] * 2
pair_tpl = [k for _ in range(T) for k in range(9)] + [9 + 9 * t + k for t in range(T) for k in range(9)]
from barc4dip_amd import _ffi
cases = (("phase", phase_correlation_batch, {}), ("ncc/skimage", template_matching_batch, dict(backend="skimage")),
         ("ncc/opencv", template_matching_batch, dict(backend="opencv")))
for lanes, (name, fn, kw) in [(l, c) for l in (1, 0, 1, 0) for c in cases]:
    _ffi.lib().b4d_set_option(b"lanes", lanes)
    res = fn(dev, dev, tpl_frame, tpl_roi, pair_img, pair_tpl, **kw); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); res = fn(dev, dev, tpl_frame, tpl_roi, pair_img, pair_tpl, **kw); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = bool(np.all(np.median(np.rint(res[:9 * T, 0]).reshape(T, 9), axis=1) == sh[:, 0]))
    print(f"lanes {lanes} {name}: {len(pair_img) / best:.0f} pairs/s, truth {ok}", flush=True)
