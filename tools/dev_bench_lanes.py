"""dev: the secondary legs of bench.py in a process that has run the headline first (like the driver's), with the option "lanes" 1 / 0
interleaved: are the two-lane routes still ahead next to 10 GB of resident stack and workspace?"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from barc4dip_amd import _ffi, synth  # noqa: E402
from barc4dip_amd.signal import psd_autocorr2d_stack  # noqa: E402

torch.cuda.set_device(0)
lib = _ffi.lib()
stack = synth.speckle_stack_device(256, 2048, seed0=1234)
for _ in range(3):
    psd_autocorr2d_stack(stack, return_tensors=True)
torch.cuda.synchronize()
for rnd in range(2):
    for lanes in (1, 0):
        assert lib.b4d_set_option(b"lanes", lanes) == 0
        f = bench.secondary_fft2d(torch, stack, False)
        c3 = bench.secondary_cfg3(torch, False)
        c5 = bench.secondary_cfg5(torch, False)
        print(f"round {rnd} lanes {lanes}: fft2d {f['frames_per_s']:.0f} frames/s  cfg3 {c3['pairs_per_s']:.0f} pairs/s  "
              f"cfg5 {c5['frames_per_s']:.0f} frames/s", flush=True)
lib.b4d_set_option(b"lanes", 1)
