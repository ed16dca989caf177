"""Translation tracking on the GPU -- drop-in for ``barc4dip.signal.tracking``.

``track_translation`` keeps the reference's registry/dispatcher (tracking.py:12-78): methods are
looked up by name in ``_TRACKERS`` and receive ``backend=``.  ``phase_correlation`` with
``backend="internal"`` (tracking.py:191-297) runs entirely on the device: z-scoring, zero
embedding, both forward transforms, whitened cross-power spectrum, inverse transform, |.|,
first-occurrence arg-max, peak, exact median SNR and the 3x3 Taylor step (including the
reference's swapped corrections, tracking.py:372-373).  ``phase_correlation_batch`` exposes
the batched form used for stacks: every distinct image and template is transformed once.

``template_matching`` (tracking.py:81-188) wraps cv2 / scikit-image in the reference; neither
library ships here, so it raises ImportError exactly as the reference does without them.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Literal

import numpy as np

from .. import _device as D
from .. import _ffi
from ..geometry.roi import roi_slices

_Tracker = Callable[..., tuple]
_TRACKERS: dict[str, _Tracker] = {}


def _register(method: str):
    key = method.strip().lower()

    def deco(fn):
        _TRACKERS[key] = fn
        return fn

    return deco


def track_translation(template, image, *, slices_yx=None, method: str = "phase",
                      backend: Literal["internal", "skimage"] = "internal", subpixel: bool = True,
                      eps: float = 1e-9):
    """Dispatcher for translation tracking methods -> (dy, dx, peak_value, snr)."""
    fn = _TRACKERS.get(method.strip().lower())
    if fn is None:
        raise ValueError(f"Unsupported tracking method: {method!r}. Supported: {', '.join(sorted(_TRACKERS))}")
    return fn(template, image, slices_yx=slices_yx, backend=backend, subpixel=subpixel, eps=eps)


def _as_float2d(a, *, name: str):
    if not D.is_tensor(a):
        a = np.asarray(a)
    if a.ndim != 2:
        raise ValueError(f"{name} must be a 2D array.")
    return a


@_register("template")
def template_matching(template, image, *, slices_yx=None, backend: Literal["opencv", "skimage"] = "opencv",
                      subpixel: bool = True, eps: float = 1e-9):
    """NCC template matching (reference: tracking.py:81-188): third-party back-ends only."""
    tpl = _as_float2d(template, name="template")
    img = _as_float2d(image, name="image")
    H, W = img.shape
    h, w = tpl.shape
    if h > H or w > W:
        raise ValueError(f"template shape {(h, w)} must fit inside image shape {(H, W)}")
    if slices_yx is None:
        roi_slices((H, W), (h, w), center_yx=None, clip=False)
    if backend == "opencv":
        raise ImportError("backend='opencv' requires opencv-python (cv2).")
    if backend == "skimage":
        raise ImportError("backend='skimage' requires scikit-image.")
    raise ValueError("backend must be 'opencv' or 'skimage'.")


def phase_correlation_batch(images, tpl_src, tpl_frame, tpl_roi, pair_img, pair_tpl, *, subpixel: bool = True,
                            eps: float = 1e-9, return_peak_ij: bool = False):
    """Batched phase correlation.

    images (nimg, ny, nx), tpl_src (nsrc, ny, nx): NumPy arrays or ROCm tensors (float32 used).
    tpl_frame (ntpl,), tpl_roi (ntpl, 4) = (y0, y1, x0, x1): template k is that ROI of tpl_src[tpl_frame[k]].
    pair_img, pair_tpl (npairs,): pair i correlates images[pair_img[i]] with template pair_tpl[i].
    Returns (npairs, 4) float64 rows (dy, dx, peak, snr) [, (npairs, 2) int32 arg-max indices]."""
    torch = _ffi.require_gpu()
    im, _, _ = D.to_device_f32(images, ndim=(3,))
    if tpl_src is images:
        ts = im
    else:
        ts, _, _ = D.to_device_f32(tpl_src, ndim=(3,))
    if tuple(ts.shape[1:]) != tuple(im.shape[1:]):
        raise ValueError("tpl_src frames must have the image shape.")
    nimg, ny, nx = im.shape
    tf = np.ascontiguousarray(tpl_frame, dtype=np.int32).ravel()
    tr = np.ascontiguousarray(tpl_roi, dtype=np.int32).reshape(-1, 4)
    pi = np.ascontiguousarray(pair_img, dtype=np.int32).ravel()
    pt = np.ascontiguousarray(pair_tpl, dtype=np.int32).ravel()
    if tf.size != tr.shape[0] or pi.size != pt.size:
        raise ValueError("index arrays have inconsistent lengths.")
    npairs = int(pi.size)
    pl = _ffi.get_plan(ny, nx)
    out = torch.empty((npairs, 4), dtype=torch.float64, device=im.device)
    pij = torch.empty((npairs, 2), dtype=torch.int32, device=im.device)
    as_p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _ffi.check(_ffi.lib().b4d_phase_correlation(
        pl.handle, D.ptr(im), int(nimg), D.ptr(ts), int(ts.shape[0]), as_p(tf), as_p(tr), int(tf.size),
        as_p(pi), as_p(pt), npairs, int(bool(subpixel)), float(eps), D.ptr(out), D.ptr(pij), _ffi.stream_ptr()))
    res = out.cpu().numpy()
    return (res, pij.cpu().numpy()) if return_peak_ij else res


@_register("phase")
def phase_correlation(template, image, *, slices_yx=None, backend: Literal["internal", "skimage"] = "internal",
                      subpixel: bool = True, eps: float = 1e-9):
    """Translation (dy, dx) of a template ROI inside a full frame by phase correlation
    (reference: tracking.py:191-297).  Returns (dy, dx, peak_value, snr) as Python floats."""
    torch = _ffi.require_gpu()
    tpl = _as_float2d(template, name="template")
    img = _as_float2d(image, name="image")
    H, W = img.shape
    h, w = tpl.shape
    if slices_yx is None:
        slices_yx = roi_slices((H, W), (h, w), center_yx=None, clip=False)
    sy, sx = slices_yx
    if (sy.stop - sy.start, sx.stop - sx.start) != (h, w):
        raise ValueError("ROI shape does not match target slice dimensions.")
    if backend == "skimage":
        raise ImportError("backend='skimage' requires scikit-image.")
    if backend != "internal":
        raise ValueError("backend must be 'internal' or 'skimage'.")
    timg, _, _ = D.to_device_f32(img[None], ndim=(3,))
    ttpl, _, _ = D.to_device_f32(tpl, ndim=(2,))
    canvas = torch.zeros((1, H, W), dtype=torch.float32, device=timg.device)
    canvas[0, sy, sx] = ttpl
    r = phase_correlation_batch(timg, canvas, [0], [[sy.start, sy.stop, sx.start, sx.stop]], [0], [0],
                                subpixel=subpixel, eps=eps)[0]
    return float(r[0]), float(r[1]), float(r[2]), float(r[3])
