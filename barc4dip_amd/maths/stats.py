"""1-D peak-width helpers (host side; inputs are a few thousand samples at most).

Same names and results as ``barc4dip.maths.stats`` (stats.py:9-156): widths are measured at
``fraction * peak`` with linear interpolation between the bracketing samples."""
from __future__ import annotations

import numpy as np


def _crossing(p: np.ndarray, i_above: int, i_below: int, thr: float) -> float:
    """Position where the segment between sample i_above (>= thr side) and i_below (< thr) meets thr."""
    lo, hi = (i_below, i_above) if i_below < i_above else (i_above, i_below)
    ya, yb = p[lo], p[hi]
    if yb == ya:
        return float(i_below)
    return lo + (thr - ya) / (yb - ya)


def width_at_fraction(profile, *, fraction: float = 1.0 / np.e, center_index=None):
    """Full width of a 1-D peak at `fraction` of its height -> (width_px, hit_edge) (reference: stats.py:9-89)."""
    p = np.asarray(profile, dtype=float)
    if p.ndim != 1 or p.size == 0:
        raise ValueError("profile must be a non-empty 1D array.")
    if not (0.0 < fraction < 1.0):
        raise ValueError("fraction must be in (0, 1).")
    c = int(np.argmax(p) if center_index is None else center_index)
    c = min(max(c, 0), p.size - 1)
    thr = p[c] * fraction
    under = p < thr
    lefts = np.flatnonzero(under[:c + 1])
    rights = np.flatnonzero(under[c:])
    if lefts.size == 0 or rights.size == 0:
        return float(p.size), True
    il, ir = int(lefts[-1]), c + int(rights[0])
    x_left = _crossing(p, il + 1, il, thr)
    x_right = _crossing(p, ir - 1, ir, thr)
    return float(x_right - x_left), False


def distance_at_fraction_from_peak(profile, *, fraction: float = 1.0 / np.e, peak_index: int = 0):
    """One-sided distance from `peak_index` to the first sample below `fraction * peak`
    -> (dist_px, hit_edge) (reference: stats.py:92-156)."""
    p = np.asarray(profile, dtype=float)
    if p.ndim != 1 or p.size == 0:
        raise ValueError("profile must be a non-empty 1D array.")
    if not (0.0 < fraction < 1.0):
        raise ValueError("fraction must be in (0, 1).")
    k0 = min(max(int(peak_index), 0), p.size - 1)
    thr = p[k0] * fraction
    under = np.flatnonzero(p[k0:] < thr)
    if under.size == 0:
        return float(p.size), True
    ic = k0 + int(under[0])
    if ic == k0:
        return 0.0, False
    return float(_crossing(p, ic - 1, ic, thr) - k0), False
