"""GPU drop-in for ``barc4dip.preprocessing.normalize.flat_field_correction`` (normalize.py:12-145).

Same signature, defaults, return type (float32 ndarray of the input shape) and ValueErrors as the reference.  All
arithmetic runs in the kernels of csrc/b4d_prep.hip in the reference's float32 operation order: the "flat_median" and
"none" scalings are bit-exact against NumPy; "flat_mean" accumulates the mean of the valid denominators in float64
(NumPy: pairwise float32), so its scale factor may differ in the last float32 bit.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _device as D
from .. import _ffi


def _upload(a, who: str, ndims=(2, 3)):
    """float32 device tensor of `a` (images.astype(np.float32), normalize.py:79/88-91)."""
    torch = _ffi.require_gpu()
    if D.is_tensor(a):
        if a.ndim not in ndims:
            raise ValueError(f"{who} must be 2D or 3D")
        return a.to(device="cuda", dtype=torch.float32).contiguous()
    arr = np.asarray(a)
    if arr.ndim not in ndims:
        raise ValueError(f"{who} must be 2D or 3D")
    return torch.from_numpy(np.ascontiguousarray(arr.astype(np.float32, copy=False))).to("cuda")


def _reduce_stack(arr):
    """2-D float32 device frame: a 3-D stack is averaged along axis 0 in float32 (normalize.py:86-93)."""
    if arr is None:
        return None
    torch = _ffi.require_gpu()
    t = _upload(arr, "flats/darks")
    if t.ndim == 2:
        return t
    n, h, w = (int(v) for v in t.shape)
    out = torch.empty((h, w), dtype=torch.float32, device=t.device)
    _ffi.check(_ffi.lib().b4d_stack_mean_f32(D.ptr(t), n, h * w, D.ptr(out), _ffi.stream_ptr()))
    return out


def _median_f32(t):
    """(np.median of the non-NaN float32 values of a device array, their count), the median as NumPy computes it for
    float32 input: the middle order statistic, or the float32 mean of the two middle ones.  NaN when nothing is left."""
    torch = _ffi.require_gpu()
    q = np.array([50.0], dtype=np.float64)
    out = torch.empty((1, 1, 4), dtype=torch.float64, device=t.device)
    _ffi.check(_ffi.lib().b4d_percentiles(D.ptr(t), 1, int(t.numel()), q.ctypes.data_as(C.c_void_p), 1, D.ptr(out), _ffi.stream_ptr()))
    lo, hi, _, n = out.cpu().numpy()[0, 0]
    if n < 1:
        return np.float32(np.nan), 0
    if int(n) % 2:
        return np.float32(lo), int(n)
    return np.float32(np.float32(lo) + np.float32(hi)) / np.float32(2), int(n)


def _mean_valid(t) -> np.float32:
    """float32(mean of the non-NaN values), accumulated in float64 by b4d_moments."""
    torch = _ffi.require_gpu()
    flat = t.reshape(-1)
    if flat.numel() % 4:    # b4d_moments reads float4: pad with NaN (skipped)
        flat = torch.cat([flat, torch.full((4 - flat.numel() % 4,), float("nan"), device=t.device)])
    out = torch.empty((1, 8), dtype=torch.float64, device=t.device)
    _ffi.check(_ffi.lib().b4d_moments(D.ptr(flat), 1, int(flat.numel()), 0.0, float("inf"), D.ptr(out), _ffi.stream_ptr()))
    n, mean = out.cpu().numpy()[0, :2]
    return np.float32(mean) if n >= 1 else np.float32(np.nan)


def flat_field_correction(images, *, flats=None, darks=None, scale: str = "flat_median", bad_pixel_removal: bool = False,
                          eps: float | None = None, verbose: bool = False, return_tensors: bool = False):
    """(I - D) / (F - D) * scale_factor in float32; pixels with F - D <= eps are zeroed, or with
    ``bad_pixel_removal`` replaced by the 3x3 median of the corrected frame (reference: normalize.py:12-145)."""
    torch = _ffi.require_gpu()
    if scale not in {"none", "flat_mean", "flat_median"}:
        raise ValueError(f"Invalid scale option: {scale}")
    if (D.is_tensor(images) and images.ndim not in (2, 3)) or (not D.is_tensor(images) and np.ndim(images) not in (2, 3)):
        raise ValueError("images must be 2D or 3D")
    img = _upload(images, "images")
    is_stack = img.ndim == 3
    flat2d = _reduce_stack(flats)
    dark2d = _reduce_stack(darks)

    def done(t):
        return t if return_tensors else D.to_host(t)

    if flat2d is None and dark2d is None:
        return done(img.clone())
    h, w = int(img.shape[-2]), int(img.shape[-1])
    for name, t in (("flats", flat2d), ("darks", dark2d)):
        if t is not None and tuple(t.shape) != (h, w):
            raise ValueError(f"operands could not be broadcast together: images {tuple(img.shape)} vs {name} {tuple(t.shape)}")
    npix = h * w
    batch = int(img.shape[0]) if is_stack else 1
    out = torch.empty_like(img)
    lib = _ffi.lib()
    st = _ffi.stream_ptr()
    null = C.c_void_p(0)
    if flat2d is None:      # darks only: I - D
        _ffi.check(lib.b4d_flat_field(D.ptr(img), batch, npix, null, D.ptr(dark2d), 0.0, 1.0, 0, D.ptr(out), st))
        return done(out)
    dptr = D.ptr(dark2d) if dark2d is not None else null
    den = torch.empty((h, w), dtype=torch.float32, device=img.device)
    _ffi.check(lib.b4d_flat_den(D.ptr(flat2d), dptr, npix, 0.0, 0, D.ptr(den), st))
    med, n_ok = _median_f32(den)
    den_has_nan = n_ok < npix            # a NaN denominator makes np.median / np.mean of it NaN (and it is never "bad")
    if eps is None:         # relative threshold from the median denominator (normalize.py:109-111)
        if den_has_nan:
            med = np.float32(np.nan)
        eps_f = np.float32(1e-6 * med) if med > 0 else np.float32(1e-6)
    else:
        eps_f = np.float32(eps)
    _ffi.check(lib.b4d_flat_den(D.ptr(flat2d), dptr, npix, float(eps_f), 1, D.ptr(den), st))   # NaN where den <= eps
    s = np.float32(1.0)
    if scale != "none" and den_has_nan:
        s = np.float32(np.nan)
    elif scale == "flat_median":
        s, _ = _median_f32(den)
    elif scale == "flat_mean":
        s = _mean_valid(den)
    _ffi.check(lib.b4d_flat_field(D.ptr(img), batch, npix, D.ptr(flat2d), dptr, float(eps_f), float(s), int(scale != "none"),
                                  D.ptr(out), st))
    if bad_pixel_removal:
        # bad = den <= eps: exactly the NaNs written by the masked pass, unless F - D itself is NaN (not "bad" in the
        # reference either: NaN <= eps is False) -- take those out again
        raw = torch.empty_like(den)
        _ffi.check(lib.b4d_flat_den(D.ptr(flat2d), dptr, npix, 0.0, 0, D.ptr(raw), st))
        idx = torch.nonzero((torch.isnan(den) & ~torch.isnan(raw)).reshape(-1)).reshape(-1).contiguous()
        if idx.numel():
            _ffi.check(lib.b4d_repair_pixels(D.ptr(out), batch, h, w, D.ptr(idx), int(idx.numel()), st))
    if verbose:
        print("flat_field_correction: done")
    return done(out)
