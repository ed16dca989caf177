"""ROI helpers (host side) -- same names and behaviour as ``barc4dip.geometry.roi`` (roi.py:9-222)."""
from __future__ import annotations

import numpy as np


def odd_size(n, *, min_size: int = 3) -> int:
    """Smallest odd integer >= max(ceil(n), min_size) (reference: geometry/roi.py:9-41)."""
    if not np.isfinite(n):
        raise ValueError("n must be finite.")
    if min_size < 1:
        raise ValueError("min_size must be >= 1.")
    size = max(int(np.ceil(n)), min_size)
    return size | 1


def roi_slices(image_shape, size_yx, *, center_yx=None, clip: bool = False):
    """(slice_y, slice_x) of an odd-sized ROI around a centre (reference: geometry/roi.py:44-106)."""
    H, W = image_shape
    size_y, size_x = size_yx
    if size_y <= 0 or size_x <= 0:
        raise ValueError("ROI sizes must be positive.")
    if size_y % 2 == 0 or size_x % 2 == 0:
        raise ValueError("ROI sizes must be odd for symmetry.")
    cy, cx = (H // 2, W // 2) if center_yx is None else center_yx
    cy, cx = int(cy), int(cx)
    lo_y, hi_y = cy - size_y // 2, cy + size_y // 2 + 1
    lo_x, hi_x = cx - size_x // 2, cx + size_x // 2 + 1
    if clip:
        lo_y, hi_y, lo_x, hi_x = max(0, lo_y), min(H, hi_y), max(0, lo_x), min(W, hi_x)
    elif lo_y < 0 or hi_y > H or lo_x < 0 or hi_x > W:
        raise ValueError("ROI exceeds image bounds.")
    return slice(lo_y, hi_y), slice(lo_x, hi_x)


_LABELS = (("NW", "N", "NE"), ("W", "C", "E"), ("SW", "S", "SE"))


def roi_grid_3x3(image_shape, roi_size_yx, step_yx, *, center_yx=None):
    """Row-major 3x3 grid of ROI slices + labels (reference: geometry/roi.py:109-172)."""
    H, W = image_shape
    if center_yx is None:
        center_yx = (H // 2, W // 2)
    sy, sx = step_yx
    grid = np.empty((3, 3), dtype=object)
    for i in range(3):
        for j in range(3):
            grid[i, j] = roi_slices(image_shape, roi_size_yx,
                                    center_yx=(int(center_yx[0] + (i - 1) * sy), int(center_yx[1] + (j - 1) * sx)),
                                    clip=False)
    return grid, np.array(_LABELS, dtype=object)


def embed_roi(roi, *, out_shape, slices_yx, fill_value: float = 0.0, dtype=None):
    """Place a 2-D ROI into a filled (H, W) canvas (reference: geometry/roi.py:175-222)."""
    sy, sx = slices_yx
    canvas = np.full(tuple(out_shape), fill_value, dtype=roi.dtype if dtype is None else dtype)
    if roi.shape != (sy.stop - sy.start, sx.stop - sx.start):
        raise ValueError("ROI shape does not match target slice dimensions.")
    canvas[sy, sx] = roi
    return canvas
