"""GPU drop-in for ``barc4dip.preprocessing`` (deconvolve_psf; flat_field_correction is SURVEY.md §8f next)."""
from __future__ import annotations

from .filters import deconvolve_psf

__all__ = ["deconvolve_psf"]
