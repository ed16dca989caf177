import sys, time, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
from barc4dip_amd import synth, _ffi
from barc4dip_amd import _device as D
st = torch.from_numpy(synth.speckle_frame(2048, 5)[None]).cuda()
lib = _ffi.lib()
qs = np.array([0.05, 99.95])
out = torch.empty((1, 2, 4), dtype=torch.float64, device="cuda")
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rc = lib.b4d_percentiles(D.ptr(st), 1, 2048 * 2048, qs.ctypes.data_as(C.c_void_p), 2, D.ptr(out), _ffi.stream_ptr())
    t1 = time.perf_counter()
    r = out.cpu().numpy(); t2 = time.perf_counter()
    print(f"call {1e3*(t1-t0):.3f} ms, d2h {1e3*(t2-t1):.3f} ms", flush=True)
from barc4dip_amd.metrics import kernels as K
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); K.percentiles_batch(st, [0.05, 99.95]); print(f"percentiles_batch {1e3*(time.perf_counter()-t0):.3f} ms")
