"""Sharpness metrics on the GPU -- drop-in for ``barc4dip.metrics.sharpness`` (sharpness.py:89-861).

Tenengrad and Laplacian variance use a fused 3x3 stencil + reduction kernel with scipy's "reflect" borders;
spectral entropy and the inverse autocorrelation widths reuse the FFT pipeline; STA2 eigenvalues run on
b4d_sta2_eigenvalues (Gram matrix on the matrix cores + block subspace iteration).  Tile policy as in
speckles.py; FFT-based tile metrics and the stack functions are batched per tile shape / over frames.
"""
from __future__ import annotations

import logging
import warnings
from typing import Literal, Sequence

import numpy as np

from .. import _device as D
from .. import _ffi
from ..signal import corr as _corr
from ..signal import fft as _fft
from . import kernels as K
from .common import (choose_tiling_mode, grids_to_fields, normalize_groups, stack_time_series,
                     tile_spans, tiles_meta)
from .speckles import (_dev2d, _fft_ok, _pad4, _pad_square_batch, _pad_square_dev, _tile_batches, _widths_batch, _widths_from_autocorr,
                       tile_batch_memo, tiled_fields_batched, tiled_fields_batched_multi)
from .statistics import distribution_moments, moments_from_sums

logger = logging.getLogger(__name__)

_SHARPNESS_UNITS: dict[str, dict[str, str]] = {
    "stats": {"mean": "a.u.", "std": "a.u.", "variance": "a.u.^2", "skewness": "", "kurtosis": "", "frac_zero": "",
              "frac_sat": "", "SNRdB": "dB"},
    "gradient": {"tenengrad": "a.u.^2", "ex": "a.u.^2", "ey": "a.u.^2", "re": ""},
    "laplacian": {"laplacian_variance": "a.u.^2"},
    "spectral": {"spectral_entropy": ""},
    "autocorrelation": {"sx": "1/px", "sy": "1/px", "seq": "1/px", "r": ""},
    "eigenvalues": {"eigenvalues": "", "e1": "", "e2": "", "re": ""},
}
_ALL_SHARPNESS_GROUPS: set[str] = {"stats", "gradient", "laplacian", "spectral", "autocorrelation", "eigenvalues"}
_FFT_GROUPS = {"spectral", "autocorrelation"}


def _check2d(image, who: str, all_finite: bool = False):
    data = image if D.is_tensor(image) else np.asarray(image)
    if data.ndim != 2:
        raise ValueError(f"Expected 2D array, got ndim={data.ndim}")
    if int(np.prod(tuple(data.shape))) == 0:
        raise ValueError(f"{who} received an empty image.")
    t = _dev2d(data)
    fin = t.isfinite()
    if all_finite:
        if not bool(fin.all()):
            raise ValueError(f"{who} requires all values to be finite.")
    elif not bool(fin.any()):
        raise ValueError(f"{who} received image with no finite values.")
    return t


def _gradient_from(row, eps: float) -> dict:
    ex, ey = float(row[0]), float(row[1])
    return {"tenengrad": float(ex + ey), "ex": ex, "ey": ey, "re": float(ex / (ey + float(eps)))}


def tenengrad(image, *, eps: float = 1e-12, verbose: bool = False) -> dict:
    """mean(gx^2) + mean(gy^2) of the Sobel gradients over finite pixels (reference: sharpness.py:405-476)."""
    t = _check2d(image, "tenengrad")
    out = _gradient_from(K.sobel_laplace_batch(t[None]).cpu().numpy()[0], eps)
    if verbose:
        logger.info("> tenengrad: %.6g | ex: %.6g | ey: %.6g | ex/ey: %.3f", out["tenengrad"], out["ex"], out["ey"], out["re"])
    return out


def laplacian_variance(image, *, verbose: bool = False) -> float:
    """Population variance of the 5-point Laplacian over finite pixels (reference: sharpness.py:482-530)."""
    t = _check2d(image, "laplacian_variance")
    row = K.sobel_laplace_batch(t[None]).cpu().numpy()[0]
    var = float(row[3] - row[2] * row[2])
    if verbose:
        logger.info("> laplacian variance: %.6g", var)
    return var


def _entropy_from(row, size: int, remove_dc: bool, p_dc: float = 0.0) -> float:
    """row: b4d_psd_stats sums over every bin but the DC one; p_dc: the DC bin's power where it takes part."""
    s_all, splnp = float(row[5]), float(row[6])
    if p_dc > 0.0:
        s_all += p_dc
        splnp += p_dc * float(np.log(p_dc))
    if not np.isfinite(s_all) or s_all <= 0.0:
        raise ValueError("PSD sum is non-positive; cannot compute spectral entropy.")
    m = int(size - 1) if remove_dc else int(size)
    if m < 2:
        raise ValueError("Insufficient number of spectral bins to compute normalized entropy.")
    h = np.log(s_all) - splnp / s_all          # -sum p ln p with p = P / S
    return float(h / np.log(float(m)))


def spectral_entropy(image, *, remove_mean: bool = True, remove_dc: bool = True, eps: float = 1e-30,
                     verbose: bool = False) -> float:
    """Normalised Shannon entropy of the PSD (reference: sharpness.py:536-629; like the reference, the image is
    NOT padded to a square).  The eps clip of the reference changes the value by < 1e-27 and is not applied."""
    t = _check2d(image, "spectral_entropy", all_finite=True)
    psd = _fft.psd2d_stack(t[None], scale=False, return_tensors=True)
    # Subtracting the mean changes the DC bin and nothing else, and the device sums leave that bin out: with the mean removed it
    # holds rounding noise (< eps after the clip: nothing), zeroed it is gone, kept with the mean in it is (sum x)^2 exactly.
    p_dc = 0.0
    if not remove_mean and not remove_dc:
        mom = K.moments_batch(_pad4(t[None]), eps=0.0, saturation=None).cpu().numpy()[0]
        p_dc = float(mom[0] * mom[1]) ** 2
    hn = _entropy_from(K.psd_stats_batch(psd)[0], int(t.numel()), remove_dc, p_dc)
    if verbose:
        logger.info("> spectral_entropy: %.6g", hn)
    return hn


def inverse_autocorr_width(image, *, fraction: float = 1.0 / np.e,
                           radial_method: Literal["binned", "interpolated"] = "interpolated", min_size_px: int = 32,
                           verbose: bool = False) -> dict:
    """sx = 1/lx, sy = 1/ly, seq = 1/leq, r = lx/ly from the standardised autocorrelation
    (reference: sharpness.py:635-746; "binned" also uses the interpolated estimator there, 704-707)."""
    data = image if D.is_tensor(image) else np.asarray(image)
    if data.ndim != 2:
        raise ValueError("image must be a 2D array.")
    if int(np.prod(tuple(data.shape))) == 0:
        raise ValueError("inverse_autocorr_width received an empty image.")
    if min(data.shape) < int(min_size_px):
        raise ValueError(f"image too small for inverse autocorrelation width (min dimension < {int(min_size_px)}).")
    if radial_method not in ("binned", "interpolated"):
        raise ValueError("radial_method must be 'binned' or 'interpolated'.")
    sq = _pad_square_dev(_dev2d(data))
    ac, _, _ = _corr.autocorr2d(sq, dx=1.0, dy=1.0, remove_mean=True, standardize=True, normalize="peak", return_tensors=True)
    lx, ly, leq = _widths_from_autocorr(ac, fraction, "interpolated")
    inv = lambda v: float(1.0 / v) if v != 0.0 else float("inf")  # noqa: E731
    out = {"sx": inv(lx), "sy": inv(ly), "seq": inv(float(leq)), "r": float(lx / ly) if ly != 0.0 else float("inf")}
    if verbose:
        logger.info("> inv_ac_width: sx=%.4g | sy=%.4g | seq=%.4g | r(lx/ly)=%.3g", out["sx"], out["sy"], out["seq"], out["r"])
    return out


def _spectral_entropy_batch(stack) -> list[dict]:
    """spectral_entropy() of every item of a (B, h, w) device stack (defaults; no padding, like the reference)."""
    if not bool(stack.isfinite().all()):
        raise ValueError("spectral_entropy requires all finite values.")
    psd = _fft.psd2d_stack(stack, scale=False, return_tensors=True)
    npx = int(stack.shape[1]) * int(stack.shape[2])
    return [{"spectral_entropy": _entropy_from(row, npx, True)} for row in K.psd_stats_batch(psd)]


def _inverse_autocorr_width_batch(stack, fraction: float = 1.0 / np.e, min_size_px: int = 32) -> list[dict]:
    """inverse_autocorr_width() of every item of a (B, h, w) device stack."""
    if min(int(stack.shape[1]), int(stack.shape[2])) < int(min_size_px):
        raise ValueError(f"image too small for inverse autocorrelation width (min dimension < {int(min_size_px)}).")
    ac = _corr.autocorr2d_stack(_pad_square_batch(stack), remove_mean=True, standardize=True, normalize="peak", return_tensors=True)
    inv = lambda v: float(1.0 / v) if v != 0.0 else float("inf")  # noqa: E731
    return [{"sx": inv(lx), "sy": inv(ly), "seq": inv(float(leq)), "r": float(lx / ly) if ly != 0.0 else float("inf")}
            for lx, ly, leq in _widths_batch(ac, fraction)]


def _sta2_device(stack, nout: int = 8) -> np.ndarray:
    """Leading STA2 eigenvalues of a (B, h, w) float32 device stack: (B, nout) float64, descending (b4d_sta2_eigenvalues)."""
    import ctypes as C

    from .. import _ffi

    b, h, w = (int(v) for v in stack.shape)
    out = np.empty((b, nout), dtype=np.float64)
    _ffi.check(_ffi.lib().b4d_sta2_eigenvalues(C.c_void_p(stack.data_ptr()), b, h, w, out.ctypes.data_as(C.c_void_p), nout,
                                                _ffi.stream_ptr()))
    return out


_STA2_MAX_K = 8         # leading eigenvalues the kernel returns


def _eig_from_svals(eig: np.ndarray, k: int, eps: float) -> dict:
    e1 = float(eig[0]) if eig.size >= 1 else 0.0
    e2 = float(eig[1]) if eig.size >= 2 else 0.0
    return {"eigenvalues": float(np.sum(eig[:min(int(k), int(eig.size))])), "e1": e1, "e2": e2, "re": float(e1 / (e2 + float(eps)))}


def _eigenvalues_batch(stack, k: int = 5, eps: float = 1e-30) -> list[dict]:
    """STA2 eigenvalues of a (B, h, w) float32 device stack, all frames in one call."""
    if int(k) > _STA2_MAX_K:
        raise NotImplementedError(f"eigenvalues: k > {_STA2_MAX_K} is not built on the GPU path (the kernel returns the leading "
                                  f"{_STA2_MAX_K} eigenvalues).")
    return [_eig_from_svals(e, k, eps) for e in _sta2_device(stack.contiguous())]


def eigenvalues(image, *, k: int = 5, eps: float = 1e-30, verbose: bool = False) -> dict:
    """STA2: eig = s^2/(M*N-1) of the energy-normalised, mean-removed image; sum of the first k, e1, e2, e1/e2
    (reference: sharpness.py:752-861).  The leading eigenvalues come from b4d_sta2_eigenvalues (Gram matrix on the
    matrix cores + block subspace iteration; a float64 Jacobi on the Gram matrix for images under 64 pixels a side);
    k <= 8."""
    import torch

    t = _check2d(image, "eigenvalues", all_finite=True)
    if int(k) < 1:
        raise ValueError("k must be >= 1.")
    if t.numel() - 1 <= 0:
        raise ValueError("eigenvalues requires at least 2 pixels (M*N >= 2).")
    out = _eigenvalues_batch(t.float().unsqueeze(0), k, eps)[0]
    if not np.isfinite(out["e1"]):       # the kernel reports frames without energy as NaN rows
        raise ValueError("eigenvalues cannot normalize an all-zero image.")
    if verbose:
        logger.info("> eigenvalues: %.6g | e1: %.6g | e2: %.6g | e1/e2: %.3f", out["eigenvalues"], out["e1"], out["e2"], out["re"])
    return out


def _tiles_pointwise_multi(tb, tile_mode, groups, saturation_value, eps) -> list[dict]:
    """Per frame of a (B, H, W) device stack: stats / gradient / laplacian / eigenvalues tile grids, every kernel launched
    once per tile shape over the tiles of all frames."""
    b = int(tb.shape[0])
    n, batches = _tile_batches(tb, tile_mode)
    st = grad = lap = eigs = None
    for _, frcs, stack in batches:
        mom = K.moments_batch(_pad4(stack), eps=eps, saturation=saturation_value).cpu().numpy() if "stats" in groups else None
        sl = K.sobel_laplace_batch(stack).cpu().numpy() if groups & {"gradient", "laplacian"} else None
        ev = _eigenvalues_batch(stack) if "eigenvalues" in groups else None
        for i, (f, r, c) in enumerate(frcs):
            if ev is not None:
                eigs = eigs or {k: np.empty((b, n, n)) for k in ev[i]}
                for k in eigs:
                    eigs[k][f, r, c] = ev[i][k]
            if mom is not None:
                d = moments_from_sums(mom[i], saturation_value)
                st = st or {k: np.empty((b, n, n)) for k in d}
                for k in st:
                    st[k][f, r, c] = d[k]
            if "gradient" in groups:
                g = _gradient_from(sl[i], 1e-12)
                grad = grad or {k: np.empty((b, n, n)) for k in g}
                for k in grad:
                    grad[k][f, r, c] = g[k]
            if "laplacian" in groups:
                lap = lap if lap is not None else {"laplacian_variance": np.empty((b, n, n))}
                lap["laplacian_variance"][f, r, c] = float(sl[i][3] - sl[i][2] * sl[i][2])
    outs = []
    for f in range(b):
        out = {}
        for name, grids in (("stats", st), ("gradient", grad), ("laplacian", lap), ("eigenvalues", eigs)):
            if grids is not None:
                out[name] = grids_to_fields({k: g[f] for k, g in grids.items()}, n)
        outs.append(out)
    return outs


def _tiles_pointwise(t, tile_mode, groups, saturation_value, eps):
    return _tiles_pointwise_multi(t[None], tile_mode, groups, saturation_value, eps)[0]


@tile_batch_memo.scoped
def sharpness_stats_batch(tb, *, groups: set, tiles: bool = True, saturation_value: float | None = 65535.0,
                          eps: float = 1e-6) -> list[dict]:
    """{"full": ..., "tiles": ...} of every frame of a (B, H, W) device stack (already in display orientation): the
    arithmetic of sharpness_stats, kernels launched once per batch / tile shape (used by sharpness_stack_stats)."""
    b, h, w = (int(v) for v in tb.shape)
    if h * w == 0:
        raise ValueError("sharpness_stats received an empty image.")
    outs = [{"full": {}} for _ in range(b)]
    fin = tb.isfinite()
    any_fin = fin.reshape(b, -1).any(dim=1)
    all_fin = bool(fin.all())
    if groups & {"stats", "gradient", "laplacian"} and not bool(any_fin.all()):
        raise ValueError("sharpness_stats received image with no finite values.")
    if "stats" in groups:
        mom = K.moments_batch(_pad4(tb), eps=eps, saturation=saturation_value).cpu().numpy()
        for f in range(b):
            outs[f]["full"]["stats"] = moments_from_sums(mom[f], saturation_value)
    if groups & {"gradient", "laplacian"}:
        sl = K.sobel_laplace_batch(tb).cpu().numpy()
        for f in range(b):
            if "gradient" in groups:
                outs[f]["full"]["gradient"] = _gradient_from(sl[f], 1e-12)
            if "laplacian" in groups:
                outs[f]["full"]["laplacian"] = {"laplacian_variance": float(sl[f][3] - sl[f][2] * sl[f][2])}
    if groups & {"spectral", "eigenvalues"} and not all_fin:
        raise ValueError("spectral_entropy / eigenvalues require all values to be finite.")
    if "spectral" in groups:
        for f, d in enumerate(_spectral_entropy_batch(tb)):
            outs[f]["full"]["spectral"] = d
    if "autocorrelation" in groups:
        for f, d in enumerate(_inverse_autocorr_width_batch(tb)):
            outs[f]["full"]["autocorrelation"] = d
    if "eigenvalues" in groups:
        if h * w - 1 <= 0:
            raise ValueError("eigenvalues requires at least 2 pixels (M*N >= 2).")
        for f, d in enumerate(_eigenvalues_batch(tb.contiguous())):
            if not np.isfinite(d["e1"]):
                raise ValueError("eigenvalues cannot normalize an all-zero image.")
            outs[f]["full"]["eigenvalues"] = d
    mode, tile_shape_px = choose_tiling_mode(h, w, tiles=tiles, min_tile_px=128)
    if mode == "off":
        return outs
    tiles_out = _tiles_pointwise_multi(tb, mode, groups, saturation_value, eps)
    fft_groups = sorted(groups & _FFT_GROUPS)
    if fft_groups:
        n, ys, xs = tile_spans(h, w, mode)
        if all(_fft_ok((y1 - y0, x1 - x0)) for y0, y1 in ys for x0, x1 in xs):
            if "spectral" in groups:
                for f, d in enumerate(tiled_fields_batched_multi(tb, mode, _spectral_entropy_batch)):
                    tiles_out[f]["spectral"] = d
            if "autocorrelation" in groups:
                for f, d in enumerate(tiled_fields_batched_multi(tb, mode, _inverse_autocorr_width_batch)):
                    tiles_out[f]["autocorrelation"] = d
        else:
            warnings.warn(f"tile statistics of {fft_groups} skipped: {tile_shape_px}-pixel tiles have no transform plan; "
                          "full-frame values are unaffected.", RuntimeWarning, stacklevel=2)
    order = ("stats", "gradient", "laplacian", "spectral", "autocorrelation", "eigenvalues")
    for f in range(b):
        if tiles_out[f]:
            outs[f]["tiles"] = {g: tiles_out[f][g] for g in order if g in tiles_out[f]}
    return outs


@tile_batch_memo.scoped
def sharpness_stats(image: np.ndarray, *, metrics: str | Sequence[str] = "all", tiles: bool = True,
                    display_origin: Literal["upper", "lower"] = "lower", saturation_value: float | None = 65535.0,
                    eps: float = 1e-6, verbose: bool = True) -> dict:
    """Sharpness metrics of one 2-D image: {"meta", "full", "tiles"} (reference: sharpness.py:89-288)."""
    if not isinstance(image, np.ndarray):
        raise TypeError("sharpness_stats expects a numpy.ndarray")
    if image.ndim != 2:
        raise ValueError(f"Expected 2D array, got ndim={image.ndim}")
    from .common import normalize_display_origin

    lower = normalize_display_origin(display_origin) == "lower"   # apply_display_origin (common.py:44-72): rows flipped before
    h, w = image.shape                                            # the metrics -- on the device (a strided host copy costs 1 ms)
    groups = normalize_groups(metrics, all_groups=_ALL_SHARPNESS_GROUPS, context="sharpness", param_name="metrics")
    if verbose:
        logger.info("\nsharpness stats for a (h x w: %.0f x %.0f) image:", h, w)
    out: dict = {"meta": {"kind": "sharpness", "display_origin": display_origin, "input_shape": (int(h), int(w)),
                          "requested_groups": sorted(groups), "units": _SHARPNESS_UNITS}, "full": {}}
    t = _dev2d(np.ascontiguousarray(image))
    if lower:
        t = t.flip(0).contiguous()
    if "stats" in groups:
        out["full"]["stats"] = distribution_moments(t, saturation_value=saturation_value, eps=eps, verbose=verbose)
    if "gradient" in groups:
        out["full"]["gradient"] = tenengrad(t, verbose=verbose)
    if "laplacian" in groups:
        out["full"]["laplacian"] = {"laplacian_variance": laplacian_variance(t, verbose=verbose)}
    if "spectral" in groups:
        out["full"]["spectral"] = {"spectral_entropy": spectral_entropy(t, verbose=verbose)}
    if "autocorrelation" in groups:
        out["full"]["autocorrelation"] = inverse_autocorr_width(t, verbose=verbose)
    if "eigenvalues" in groups:
        out["full"]["eigenvalues"] = eigenvalues(t, verbose=verbose)

    mode, tile_shape_px = choose_tiling_mode(h, w, tiles=tiles, min_tile_px=128)
    if mode == "off":
        return out
    out["meta"].update(tiles_meta(h, w, tile_mode=mode, tile_shape_px=tile_shape_px))
    tiles_out = _tiles_pointwise(t, mode, groups, saturation_value, eps)
    fft_groups = sorted(groups & _FFT_GROUPS)
    if fft_groups:
        n, ys, xs = tile_spans(h, w, mode)
        if all(_fft_ok((y1 - y0, x1 - x0)) for y0, y1 in ys for x0, x1 in xs):
            if "spectral" in groups:
                tiles_out["spectral"] = tiled_fields_batched(t, mode, _spectral_entropy_batch)
            if "autocorrelation" in groups:
                tiles_out["autocorrelation"] = tiled_fields_batched(t, mode, _inverse_autocorr_width_batch)
        else:
            warnings.warn(f"tile statistics of {fft_groups} skipped: {tile_shape_px}-pixel tiles need a general-length "
                          "FFT plan (not built yet); full-frame values are unaffected.", RuntimeWarning, stacklevel=2)
    if tiles_out:
        order = ("stats", "gradient", "laplacian", "spectral", "autocorrelation", "eigenvalues")
        out["tiles"] = {g: tiles_out[g] for g in order if g in tiles_out}
    return out


def sharpness_stack_stats(stack: np.ndarray, *, metrics: str | Sequence[str] = "all", tiles: bool = True,
                          display_origin: Literal["upper", "lower"] = "lower", saturation_value: float | None = 65535.0,
                          eps: float = 1e-6, verbose: bool = True, parallel: bool = True, n_jobs: int | None = None) -> dict:
    """Per-frame sharpness metrics stacked along T (reference: sharpness.py:290-399)."""
    if not isinstance(stack, np.ndarray):
        raise TypeError("sharpness_stack_stats expects a numpy.ndarray")
    if stack.ndim != 3:
        raise ValueError(f"stack must be a 3D array with shape (T, H, W); got ndim={stack.ndim}")
    T, H, W = (int(v) for v in stack.shape)
    if T < 1:
        raise ValueError("stack must contain at least one frame.")
    groups = normalize_groups(metrics, all_groups=_ALL_SHARPNESS_GROUPS, context="sharpness", param_name="metrics")
    tile_mode, tile_shape_px = choose_tiling_mode(H, W, tiles=tiles)
    from .common import normalize_display_origin

    lower = normalize_display_origin(display_origin) == "lower"
    dev_all, _, _ = D.to_device_f32(stack, ndim=(3,))
    fb = max(1, min(T, (256 << 20) // (4 * H * W)))     # frames per batch: <= 256 MiB of pixels
    per_frame = []
    for a in range(0, T, fb):
        tb = dev_all[a:a + fb]
        if lower:                                       # apply_display_origin: rows flipped before the metrics
            tb = tb.flip(1).contiguous()
        per_frame.extend(sharpness_stats_batch(tb, groups=groups, tiles=tiles, saturation_value=saturation_value, eps=eps))
    meta = {"kind": "sharpness_stack_stats", "input_shape": (H, W), "stack_shape": (T, H, W), "n_frames": T,
            "display_origin": display_origin, "requested_groups": sorted(groups), "units": _SHARPNESS_UNITS,
            "parallel": {"enabled": False, "n_jobs": None}}
    meta.update(tiles_meta(H, W, tile_mode=tile_mode, tile_shape_px=tile_shape_px))
    out = {"meta": meta, "full": stack_time_series([d["full"] for d in per_frame])}
    if tiles and all(isinstance(d.get("tiles"), dict) for d in per_frame):
        out["tiles"] = stack_time_series([d["tiles"] for d in per_frame])
    if verbose:
        logger.info("> sharpness_stack_stats | frames=%d | device batch", T)
    return out
