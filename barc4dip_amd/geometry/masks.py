"""Symmetric padding to a square (reference: geometry/masks.py:11-57)."""
from __future__ import annotations

import numpy as np

from .roi import embed_roi


def pad_to_square(image: np.ndarray, *, fill_value: float = 0.0, dtype=None) -> np.ndarray:
    if image.ndim != 2:
        raise ValueError("Input must be a 2D array.")
    H, W = image.shape
    N = max(H, W)
    y0, x0 = (N - H) // 2, (N - W) // 2
    return embed_roi(image, out_shape=(N, N), slices_yx=(slice(y0, y0 + H), slice(x0, x0 + W)),
                     fill_value=fill_value, dtype=dtype)
