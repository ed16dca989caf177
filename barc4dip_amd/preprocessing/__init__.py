"""GPU drop-in for ``barc4dip.preprocessing`` (deconvolve_psf, flat_field_correction)."""
from __future__ import annotations

from .filters import deconvolve_psf
from .normalize import flat_field_correction

__all__ = ["deconvolve_psf", "flat_field_correction"]
