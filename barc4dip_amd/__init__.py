"""barc4dip_amd -- MI355X-native hot path of barc4dip (signal / metrics / preprocessing).

Python host code mirrors the reference's function signatures and calls hand-written gfx950
kernels through the C ABI declared in include/b4d.h.  There is no CPU fallback.
"""
from __future__ import annotations

__version__ = "0.3.0"

from . import geometry, maths, metrics, preprocessing, signal  # noqa: F401
from .metrics import (distribution_moments, sharpness_stack_stats, sharpness_stats, speckle_stack_stats,  # noqa: F401
                      speckle_stats)


def set_option(name: str, value: int) -> None:
    """Process-wide library switches that change the route, never the results (include/b4d.h: ``b4d_set_option``), e.g.
    ``set_option("lanes", 0)`` keeps every launch on the caller's stream.  Raises ValueError for unknown names / values."""
    from . import _ffi

    if _ffi.lib().b4d_set_option(str(name).encode(), int(value)) != 0:
        raise ValueError(_ffi.lib().b4d_last_error().decode("utf-8", "replace"))
