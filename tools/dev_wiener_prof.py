"""dev: Wiener deconvolution of 2048^2 (padded 2056 = 8 * 257: Bluestein route) and 1024^2 (1032 = 8 * 3 * 43: fused route) frames for
rocprofv3 kernel statistics."""
import sys

import torch

sys.path.insert(0, ".")
from barc4dip_amd import synth  # noqa: E402
from barc4dip_amd.preprocessing import deconvolve_psf  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = synth.speckle_stack_device(16, n)
for _ in range(3):
    deconvolve_psf(dev, sigma=1.5, return_tensors=True)
torch.cuda.synchronize()
