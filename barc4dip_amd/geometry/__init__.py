from __future__ import annotations

from . import masks, roi
from .masks import pad_to_square
from .roi import embed_roi, odd_size, roi_grid_3x3, roi_slices

__all__ = ["masks", "roi", "pad_to_square", "embed_roi", "odd_size", "roi_grid_3x3", "roi_slices"]
