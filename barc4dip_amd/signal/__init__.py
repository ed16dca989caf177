"""GPU drop-in for ``barc4dip.signal`` (same public names as signal/__init__.py:6-26)."""
from __future__ import annotations

from . import corr, fft, tracking
from .corr import autocorr2d, autocorr2d_stack, psd_autocorr2d_stack, xcorr2d
from .tracking import (phase_correlation, phase_correlation_batch, template_matching, template_matching_batch,
                       track_translation)
from .fft import fft1d, fft2d, fft2d_stack, freq_axes2d, freq_axis1d, psd1d, psd2d, psd2d_stack

__all__ = [
    "fft", "corr", "tracking", "phase_correlation", "template_matching", "track_translation",
    "phase_correlation_batch", "template_matching_batch",
    "freq_axis1d", "freq_axes2d", "fft1d", "fft2d", "psd1d", "psd2d", "xcorr2d", "autocorr2d",
    "fft2d_stack", "psd2d_stack", "autocorr2d_stack", "psd_autocorr2d_stack",
]
