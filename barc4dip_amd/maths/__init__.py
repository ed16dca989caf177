from __future__ import annotations

from . import radial, stats
from .radial import radial_mean_binned, radial_mean_interpolated
from .stats import distance_at_fraction_from_peak, width_at_fraction

__all__ = ["radial", "stats", "radial_mean_binned", "radial_mean_interpolated", "width_at_fraction",
           "distance_at_fraction_from_peak"]
