"""dev: median of the |corr| maps of the cfg3 protocol (peak / snr) against the Rayleigh expectation sqrt(ln 2 / N)."""
import sys
sys.path.insert(0, ".")
import numpy as np
from barc4dip_amd import synth
from barc4dip_amd.geometry import roi_grid_3x3
from barc4dip_amd.signal import phase_correlation_batch
T, n, side = 4, 1024, 121
stack, sh = synth.shifted_stack(T, n, seed=1234, max_shift=32)
grid, _ = roi_grid_3x3((n, n), (side, side), (side // 2, side // 2))
rois = [(s[0].start, s[0].stop, s[1].start, s[1].stop) for s in grid.ravel()]
res = phase_correlation_batch(stack, stack, [0] * 9, rois, [t for t in range(T) for _ in range(9)], [k for _ in range(T) for k in range(9)])
med = res[:, 2] / res[:, 3]
print("expected", np.sqrt(np.log(2) / n / n), "median min/max", med.min(), med.max(), "peak", res[:, 2].min(), res[:, 2].max())
b = (np.float32(med).view(np.uint32) >> 21) + 1024
print("bins", np.unique(b, return_counts=True), "expected bin", (np.float32(np.sqrt(np.log(2) / n / n)).view(np.uint32) >> 21) + 1024)
