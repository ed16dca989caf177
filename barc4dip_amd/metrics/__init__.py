"""GPU drop-in for ``barc4dip.metrics`` (same public names as metrics/__init__.py:6-20) + temporal statistics."""
from __future__ import annotations

from . import kernels, sharpness, speckles, statistics, temporal
from .sharpness import sharpness_stack_stats, sharpness_stats
from .speckles import speckle_stack_stats, speckle_stats
from .statistics import distribution_moments
from . import sharded
from .temporal import temporal_stats

__all__ = ["sharpness", "sharpness_stats", "sharpness_stack_stats", "statistics", "speckles", "speckle_stats",
           "speckle_stack_stats", "distribution_moments", "temporal", "temporal_stats", "sharded", "kernels"]
