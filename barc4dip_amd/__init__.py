"""barc4dip_amd -- MI355X-native hot path of barc4dip (signal / metrics / preprocessing).

Python host code mirrors the reference's function signatures and calls hand-written gfx950
kernels through the C ABI declared in include/b4d.h.  There is no CPU fallback.
"""
from __future__ import annotations

__version__ = "0.1.0"

from . import geometry, maths, metrics, preprocessing, signal  # noqa: F401
from .metrics import (distribution_moments, sharpness_stack_stats, sharpness_stats, speckle_stack_stats,  # noqa: F401
                      speckle_stats)
