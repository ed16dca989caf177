"""dev: b4d_xcorr2d (row a4) over a stack of frame pairs against the plan's chunk, one stream and two plans on two streams taking
alternate chunks (is the two-lane recipe worth wiring into this entry point?)."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi  # noqa: E402

lib = _ffi.lib()
ny = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nx = int(sys.argv[2]) if len(sys.argv) > 2 else ny
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
chunks = [int(c) for c in sys.argv[4].split(",")] if len(sys.argv) > 4 else [2, 4, 8, 16, 64]
GEN = not (ny & (ny - 1) == 0 and nx & (nx - 1) == 0)
a = torch.rand((T, ny, nx), device="cuda") * 1000
b = torch.rand((T, ny, nx), device="cuda") * 1000
out = torch.empty((T, ny, nx), dtype=torch.float32, device="cuda")
fpix = ny * nx
s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
flags = _ffi.REMOVE_MEAN | _ffi.NORM_PEAK


def timed(f):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        f()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def call(pl, b0, nb, stream):
    _ffi.check(lib.b4d_xcorr2d(pl.handle, C.c_void_p(a.data_ptr() + 4 * fpix * b0), C.c_void_p(b.data_ptr() + 4 * fpix * b0), nb,
                               C.c_void_p(out.data_ptr() + 4 * fpix * b0), flags, stream))


for rep in range(2):
    for chunk in chunks:
        if chunk > T:
            continue
        pl = _ffi.Plan(ny, nx, chunk, general=GEN)
        best = timed(lambda: call(pl, 0, T, _ffi.stream_ptr()))
        line = f"{ny}x{nx} chunk {chunk:4d}: {T / best:9.0f} pairs/s"
        if 2 * chunk <= T:
            pl2 = _ffi.Plan(ny, nx, chunk, general=GEN)
            pls = (pl, pl2)

            def duo():
                for i, b0 in enumerate(range(0, T, chunk)):
                    call(pls[i & 1], b0, min(chunk, T - b0), C.c_void_p(s2[i & 1].cuda_stream))
            bd = timed(duo)
            line += f"   duo: {T / bd:9.0f} pairs/s"
            pl2.close()
        print(line, flush=True)
        pl.close()
