"""dev: does a small launch group keep fft2d's half-spectrum workspace in the 256-MB memory-side cache?  frames/s of b4d_fft2d over a
stack against the plan's chunk (frames per launch pair); `exp` = b4d_set_option("exp", v); `duo` = two plans on two streams taking
alternate chunks (column pass of one chunk under the row pass of the other)."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from barc4dip_amd import _ffi  # noqa: E402

lib = _ffi.lib()
ny = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
nx = int(sys.argv[2]) if len(sys.argv) > 2 else ny
T = int(sys.argv[3]) if len(sys.argv) > 3 else 256
chunks = [int(c) for c in sys.argv[4].split(",")] if len(sys.argv) > 4 else [1, 2, 4, 8, 16, 32, 64, 256]
exps = [int(c) for c in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
st = torch.rand((T, ny, nx), device="cuda") * 1000
out = torch.empty((T, ny, nx), dtype=torch.complex64, device="cuda")
fpix = ny * nx
GEN = not (ny & (ny - 1) == 0 and nx & (nx - 1) == 0)
s2 = [torch.cuda.Stream(), torch.cuda.Stream()]


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


for rep in range(2):
    for exp in exps:
        _ffi.check(lib.b4d_set_option(b"exp", exp))
        for chunk in chunks:
            if chunk > T:
                continue
            pl = _ffi.Plan(ny, nx, chunk, general=GEN)
            best = timed(lambda: _ffi.check(lib.b4d_fft2d(pl.handle, C.c_void_p(st.data_ptr()), T, C.c_void_p(out.data_ptr()), _ffi.stream_ptr())))
            line = (f"{ny}x{nx} exp {exp} chunk {chunk:4d} ({chunk * ny * (nx // 2) * 8 / 2**20:7.0f} MB): {T / best:9.0f} frames/s "
                    f"{12 * fpix * T / best / 8e12:.3f}")
            if 2 * chunk <= T:
                pl2 = _ffi.Plan(ny, nx, chunk, general=GEN)
                pls = (pl, pl2)

                def duo():
                    for i, b0 in enumerate(range(0, T, chunk)):
                        nb = min(chunk, T - b0)
                        _ffi.check(lib.b4d_fft2d(pls[i & 1].handle, C.c_void_p(st.data_ptr() + 4 * fpix * b0), nb,
                                                 C.c_void_p(out.data_ptr() + 8 * fpix * b0), C.c_void_p(s2[i & 1].cuda_stream)))
                bd = timed(duo)
                line += f"   duo: {T / bd:9.0f} frames/s {12 * fpix * T / bd / 8e12:.3f}"
                pl2.close()
            print(line, flush=True)
            pl.close()
_ffi.check(lib.b4d_set_option(b"exp", 0))
