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


def _crossings_batch(p: np.ndarray, rows: np.ndarray, i_above: np.ndarray, i_below: np.ndarray, thr: np.ndarray) -> np.ndarray:
    """_crossing for one (above, below) sample pair per row of p."""
    lo, hi = np.minimum(i_above, i_below), np.maximum(i_above, i_below)
    ya, yb = p[rows, lo], p[rows, hi]
    flat = yb == ya
    with np.errstate(divide="ignore", invalid="ignore"):
        x = lo + (thr - ya) / (yb - ya)
    return np.where(flat, i_below.astype(float), x)


def widths_at_fraction_batch(profiles, centers, *, fraction: float = 1.0 / np.e):
    """width_at_fraction for every row of a (B, n) array with its own centre index -> (widths (B,), hit_edge (B,)).
    Same arithmetic, one pass of array operations instead of B Python calls (the tile aggregators measure ~160 peaks a frame).
    Equal to the scalar function wherever that one is meaningful (checked on random profiles incl. ties and flats); for a
    centre value <= 0 the scalar code indexes p[-1] or past the end (a negative "width" or an IndexError) -- here indices clamp."""
    p = np.asarray(profiles, dtype=float)
    if p.ndim != 2 or p.shape[1] == 0:
        raise ValueError("profiles must be a non-empty (B, n) array.")
    if not (0.0 < fraction < 1.0):
        raise ValueError("fraction must be in (0, 1).")
    b, n = p.shape
    rows, ar = np.arange(b), np.arange(n)[None, :]
    c = np.clip(np.asarray(centers, dtype=np.int64), 0, n - 1)
    thr = p[rows, c] * fraction
    under = p < thr[:, None]
    il = np.where(under & (ar <= c[:, None]), ar, -1).max(axis=1)      # last sample under the threshold at or left of the centre
    ir = np.where(under & (ar >= c[:, None]), ar, n).min(axis=1)       # first one at or right of it
    edge = (il < 0) | (ir >= n)
    il_s, ir_s = np.clip(il, 0, n - 2), np.clip(ir, 1, n - 1)
    w = _crossings_batch(p, rows, ir_s - 1, ir_s, thr) - _crossings_batch(p, rows, il_s + 1, il_s, thr)
    return np.where(edge, float(n), w), edge


def distances_at_fraction_from_peak_batch(profiles, *, fraction: float = 1.0 / np.e, peak_index: int = 0):
    """distance_at_fraction_from_peak for every row of a (B, n) array -> (distances (B,), hit_edge (B,))."""
    p = np.asarray(profiles, dtype=float)
    if p.ndim != 2 or p.shape[1] == 0:
        raise ValueError("profiles must be a non-empty (B, n) array.")
    if not (0.0 < fraction < 1.0):
        raise ValueError("fraction must be in (0, 1).")
    b, n = p.shape
    rows, ar = np.arange(b), np.arange(n)[None, :]
    k0 = min(max(int(peak_index), 0), n - 1)
    thr = p[:, k0] * fraction
    ic = np.where((p < thr[:, None]) & (ar >= k0), ar, n).min(axis=1)
    edge = ic >= n
    ic_s = np.clip(ic, 1, n - 1)
    d = _crossings_batch(p, rows, ic_s - 1, ic_s, thr) - k0
    d = np.where(ic == k0, 0.0, d)
    return np.where(edge, float(n), d), edge
