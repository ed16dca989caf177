"""dev: randomised size sweep of the transform entry points against NumPy (soak test; not part of the suite)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from barc4dip_amd import _ffi, signal as gs

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
t0 = time.time()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    HI = int(sys.argv[3]) if len(sys.argv) > 3 else 1400
    ny, nx = int(rng.integers(2, HI)), int(rng.integers(2, HI))
    if rng.random() < 0.2:
        ny = int(2 ** rng.integers(6, 11))
    if not _ffi.supported(ny, nx):
        print("unsupported", ny, nx); continue
    img = (rng.random((ny, nx)) * 100 + 1).astype(np.float32)
    r = img.astype(np.float64)
    F = gs.fft2d(img)[0]; Fr = np.fft.fftshift(np.fft.fft2(r))
    e1 = np.max(np.abs(F - Fr)) / np.max(np.abs(Fr))
    P = gs.psd2d(img, scale=False)[0]; Pr = np.abs(Fr) ** 2
    e2 = np.max(np.abs(P - Pr)) / np.max(Pr)
    ac = gs.autocorr2d(img)[0]
    a = r - r.mean(); cr = np.fft.fftshift(np.fft.ifft2(np.abs(np.fft.fft2(a)) ** 2)).real; cr /= np.max(np.abs(cr))
    e3 = np.max(np.abs(ac - cr))
    ok = e1 < 1e-5 and e2 < 1e-5 and e3 < 1e-5 and ac[ny // 2, nx // 2] == 1.0
    if not ok:
        bad += 1
        print("FAIL", ny, nx, e1, e2, e3, ac[ny // 2, nx // 2], flush=True)
print(f"soak done: {it + 1} sizes, {bad} failures, {time.time() - t0:.0f} s", flush=True)
