"""Intensity distribution moments on the GPU -- drop-in for ``barc4dip.metrics.statistics``."""
from __future__ import annotations

import logging

import numpy as np

from . import kernels as K

logger = logging.getLogger(__name__)


def moments_from_sums(row, saturation_value) -> dict:
    """{n_finite, mean, sum d^2, sum d^3, sum d^4, n_zero, n_sat} -> the reference's dict (statistics.py:75-105)."""
    n, mean, m2, m3, m4, nz, ns = (float(v) for v in row[:7])
    if n <= 0:
        raise ValueError("distribution_moments received no finite values.")
    var = m2 / n
    std = float(np.sqrt(var))
    with np.errstate(divide="ignore", invalid="ignore"):
        skew = float((m3 / n) / var ** 1.5) if var > 0 else float("nan")
        kurt = float((m4 / n) / (var * var) - 3.0) if var > 0 else float("nan")
    if std == 0.0:
        snr_db = float("inf") if mean > 0.0 else float("nan")
    else:
        ratio = mean / std
        snr_db = float(20.0 * np.log10(ratio)) if ratio > 0.0 else (float("-inf") if ratio == 0.0 else float("nan"))
    return {"mean": mean, "std": std, "variance": float(std * std), "skewness": skew, "kurtosis": kurt,
            "frac_zero": nz / n, "frac_sat": float("nan") if saturation_value is None else ns / n, "SNRdB": snr_db}


def distribution_moments(image, *, saturation_value: float | None = 65535.0, eps: float = 1e-6, verbose: bool = False) -> dict:
    """mean / std / variance / biased skewness / Fisher kurtosis / zero and saturation fractions / SNR in dB of
    the finite pixels (reference: statistics.py:17-125)."""
    from .. import _device as D

    data = image if D.is_tensor(image) else np.asarray(image)
    if data.ndim not in (1, 2):
        raise ValueError(f"Expected 1D or 2D array, got ndim={data.ndim}")
    if int(np.prod(tuple(data.shape))) == 0:
        raise ValueError("distribution_moments received an empty image.")
    flat = data.reshape(1, -1)
    pad = (-flat.shape[1]) % 4
    if pad:  # the kernel reads 16-byte vectors: pad with NaN (ignored as non-finite)
        if D.is_tensor(flat):
            import torch

            flat = torch.cat([flat.float(), torch.full((1, pad), float("nan"), device=flat.device)], dim=1)
        else:
            flat = np.concatenate([np.asarray(flat, dtype=np.float32), np.full((1, pad), np.nan, dtype=np.float32)], axis=1)
    sums = K.moments_batch(flat, eps=eps, saturation=saturation_value).cpu().numpy()[0]
    out = moments_from_sums(sums, saturation_value)
    if verbose:
        logger.info("> moments: mean=%.0f | std=%.0f | var=%.0f | skew=%.2f | kurt=%.2f | SNR=%.2f dB | zero=%.6f | sat=%.6f",
                    out["mean"], out["std"], out["variance"], out["skewness"], out["kurtosis"], out["SNRdB"],
                    out["frac_zero"], out["frac_sat"])
    return out
