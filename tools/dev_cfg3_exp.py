"""dev: cfg3 leg of bench.py under b4d_set_option("exp", v) for v in argv (interleaved, 3 rounds); optional first argument: a
library build to load instead of the shipped one."""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from barc4dip_amd import _ffi  # noqa: E402

args = sys.argv[1:]
if args and args[0].endswith(".so"):      # another build of the library for this process
    _ffi._lib = _ffi.load_library(args.pop(0))
lib = _ffi.lib()
torch.cuda.set_device(0)
vs = [int(a) for a in args] or [0, 1]
for rnd in range(3):
    for v in vs:
        assert lib.b4d_set_option(b"exp", v) == 0
        r = bench.secondary_cfg3(torch, False)
        print(f"round {rnd} exp {v}: {r['pairs_per_s']:.0f} pairs/s truth {r['ground_truth_recovered']}", flush=True)
lib.b4d_set_option(b"exp", 0)
