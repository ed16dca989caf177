import sys, time, warnings
sys.path.insert(0, ".")
import numpy as np, torch
from barc4dip_amd import metrics as gm, synth
warnings.simplefilter("ignore")
img = synth.speckle_frame(2048, 1234)
gm.sharpness_stats(img, metrics=("eigenvalues",), verbose=False); torch.cuda.synchronize()
t0 = time.perf_counter(); gm.sharpness_stats(img, metrics=("eigenvalues",), verbose=False); torch.cuda.synchronize()
print("eigenvalues group: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
