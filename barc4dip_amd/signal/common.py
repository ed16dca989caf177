"""Calibration of sampling axes for the host side of fft.py / corr.py.

Contract taken from the reference's argument rules (signal/common.py:13-90): a transform is calibrated EITHER by pixel
steps (dx, dy > 0) OR by explicit, uniformly sampled coordinate axes, never by both; every violation is a ValueError.
"""
from __future__ import annotations

import numpy as np

_UNIFORMITY_TOL = 1e-6      # largest relative spread of the sample spacing still called "uniform"


def _axis_step(axis, expected_size: int | None, label: str) -> float:
    """Spacing of an explicit coordinate axis after checking shape, length, finiteness, monotony and uniformity."""
    coords = np.asarray(axis, dtype=float)
    problems = None
    if coords.ndim != 1:
        problems = "be one-dimensional"
    elif coords.size < 2:
        problems = "hold at least two samples"
    elif expected_size is not None and coords.size != expected_size:
        problems = f"have {expected_size} samples, one per pixel"
    if problems:
        raise ValueError(f"axis {label!r} must {problems}.")
    gaps = np.diff(coords)
    if not np.isfinite(gaps).all():
        raise ValueError(f"axis {label!r} has NaN or infinite coordinates.")
    if not ((gaps > 0).all() or (gaps < 0).all()):
        raise ValueError(f"axis {label!r} is not strictly monotonic.")
    spacing = np.abs(gaps)
    typical = float(np.median(spacing))
    spread = float(np.abs(spacing - typical).max() / typical) if typical > 0 else np.inf
    if not spread <= _UNIFORMITY_TOL:
        raise ValueError(f"axis {label!r} is not uniformly sampled (relative spread {spread:.1e}).")
    return typical


def _pick_step(axis, step: float, expected_size: int | None, label: str) -> float:
    """One axis: the explicit coordinates win if given (then the step must be left at its default)."""
    if axis is None:
        if not step > 0:
            raise ValueError(f"the pixel step d{label} must be positive.")
        return float(step)
    if step != 1.0:
        raise ValueError(f"calibrate axis {label!r} with coordinates or with d{label}, not with both.")
    return _axis_step(axis, expected_size, label)


def _resolve_step_1d(*, n: int, x, dx: float, name: str) -> float:
    return _pick_step(x, dx, int(n), name)


def _resolve_steps_2d(*, shape, x, y, dx: float, dy: float) -> tuple[float, float]:
    rows, cols = (int(v) for v in shape)
    if (x is None) ^ (y is None):
        raise ValueError("explicit axes come in pairs: give x and y together, or neither.")
    return _pick_step(x, dx, cols, "x"), _pick_step(y, dy, rows, "y")


def _lag_axis_from_step(n: int, step: float) -> np.ndarray:
    """Lag coordinates centred on n // 2 (zero lag at the fftshift-ed centre)."""
    centre = n // 2
    return float(step) * (np.arange(n, dtype=float) - centre)
