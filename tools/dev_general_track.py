"""dev: phase correlation / xcorr2d on a general (non power-of-two) detector format, timed (tools/, not product)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from barc4dip_amd.signal import phase_correlation_batch, xcorr2d

ny, nx, T = 2160, 2560, 8
rng = np.random.default_rng(3)
base = rng.poisson(300.0, size=(ny, nx)).astype(np.float32)
stack = np.stack([np.roll(base, (3 * t, -2 * t), axis=(0, 1)) + rng.poisson(5.0, size=(ny, nx)).astype(np.float32) for t in range(T)])
dev = torch.from_numpy(stack).cuda()
roi = [(900, 1261, 1100, 1461)]
tpl_frame, tpl_roi = [0], roi
pair_img, pair_tpl = list(range(T)), [0] * T
res = phase_correlation_batch(dev, dev, tpl_frame, tpl_roi, pair_img, pair_tpl)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    res = phase_correlation_batch(dev, dev, tpl_frame, tpl_roi, pair_img, pair_tpl)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"phase correlation {ny}x{nx}: {T / dt:.0f} pairs/s; shifts", np.rint(res[:, :2]).astype(int).tolist()[:4])
xcorr2d(stack[0], stack[1])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    xcorr2d(dev[0], dev[1], return_tensors=True)
torch.cuda.synchronize()
print(f"xcorr2d {ny}x{nx}: {3 / (time.perf_counter() - t0):.1f} calls/s")
