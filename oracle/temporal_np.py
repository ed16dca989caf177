"""Oracle (test infrastructure, NOT product code): per-pixel temporal statistics
of a (T, H, W) stack (BASELINE.json config 4).

The reference has no per-pixel temporal variance function; its temporal
reductions are ``data.mean(axis=0)`` (io/rw.py:129-132) and
``arr.astype(float32).mean(axis=0)`` (preprocessing/normalize.py:86-93).  The
build defines, consistently with the package's ddof=0 convention
(metrics/statistics.py:76): mean_t = sum(x)/T, var_t = sum((x-mean_t)^2)/T,
contrast_t = sqrt(var_t)/mean_t.  The oracle is those NumPy expressions in
float64.
"""
from __future__ import annotations

import numpy as np


def temporal_stats(stack):
    """Return (mean_t, var_t, contrast_t), each (H, W) float64."""
    s = np.asarray(stack)
    if s.ndim != 3:
        raise ValueError(f"stack must be (T, H, W); got ndim={s.ndim}")
    if s.shape[0] < 1:
        raise ValueError("stack must contain at least one frame.")
    s64 = s.astype(np.float64, copy=False)
    mean = s64.mean(axis=0)
    var = s64.var(axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        contrast = np.sqrt(var) / mean
    return mean, var, contrast


def temporal_sums(stack):
    """Partial sums (sum x, sum x^2, T) in float64 -- what one rank contributes to the all-reduce."""
    s64 = np.asarray(stack).astype(np.float64, copy=False)
    return s64.sum(axis=0), (s64 * s64).sum(axis=0), int(s64.shape[0])


def finalize_sums(sx, sxx, count):
    """mean/var/contrast from reduced sums: var = sxx/T - mean^2 (float64)."""
    mean = sx / float(count)
    var = np.maximum(sxx / float(count) - mean * mean, 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        contrast = np.sqrt(var) / mean
    return mean, var, contrast


def stack_mean_f32(stack):
    """io/rw.py:129-132 / normalize.py:86-93 -- float32 mean over the frame axis."""
    return np.asarray(stack).astype(np.float32).mean(axis=0)
