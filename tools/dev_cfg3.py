"""dev: cfg3 (phase-correlation tracking, 3x3 grid protocol) with a given library build."""
import sys
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from barc4dip_amd import _ffi
if len(sys.argv) > 1:
    _ffi._lib = _ffi.load_library(sys.argv[1])
import bench_configs
bench_configs.cfg3()
