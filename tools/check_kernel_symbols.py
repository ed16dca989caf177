#!/usr/bin/env python3
"""One symbol, one code: fail if a kernel symbol is defined in the DEVICE code objects of two translation units that were compiled
with different flags -- both fat binaries would carry code for one name, and which one a launch runs would be decided by
registration order, not by the caller (b4d_fft2d.hpp: B4D_UNIT_TAG / B4D_UNIT_PASSES).

The gfx950 code object of each host object is taken out of its .hip_fatbin section (llvm-objcopy, clang-offload-bundler); a kernel
is a symbol with a kernel descriptor (`<name>.kd`).

usage: check_kernel_symbols.py <obj>=<flags> ...      (csrc/Makefile: `make check-symbols`, also run by the library target)"""
import os
import subprocess
import sys
import tempfile
from collections import defaultdict

LLVM = os.environ.get("B4D_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def device_kernels(obj: str, tmp: str):
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(tmp, "unused.o")], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--targets={TARGET}", f"--input={fat}",
                    f"--output={co}"], check=True)
    out = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "-W", co], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1][:-3] for ln in out.splitlines() if ln.rstrip().endswith(".kd")}


def main():
    objs = dict(a.split("=", 1) for a in sys.argv[1:])
    where = defaultdict(list)
    with tempfile.TemporaryDirectory() as tmp:
        for obj in objs:
            for k in device_kernels(obj, tmp):
                where[k].append(obj)
    bad = {k: v for k, v in where.items() if len({objs[o].strip() for o in v}) > 1}
    for k, v in sorted(bad.items()):
        print(f"kernel compiled with different flags in {', '.join(v)}: {k}", file=sys.stderr)
    print(f"check-symbols: {len(where)} kernel symbols in {len(objs)} device code objects, {len(bad)} defined under different flags")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
