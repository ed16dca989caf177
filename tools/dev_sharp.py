import sys, time, warnings
import torch
sys.path.insert(0, ".")
from barc4dip_amd import metrics as gm, synth
warnings.simplefilter("ignore")
img = synth.speckle_frame(2048, 1234)
for g in ("stats", "gradient", "laplacian", "spectral", "autocorrelation", "eigenvalues"):
    for tiles in (False, True):
        gm.sharpness_stats(img, metrics=g, tiles=tiles, verbose=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        gm.sharpness_stats(img, metrics=g, tiles=tiles, verbose=False)
        torch.cuda.synchronize(); print(g, "tiles" if tiles else "full", round(time.perf_counter() - t0, 4), flush=True)
