from __future__ import annotations

from . import kernels, temporal
from .temporal import temporal_stats

__all__ = ["kernels", "temporal", "temporal_stats"]
