"""Speckle-field metrics on the GPU -- drop-in for ``barc4dip.metrics.speckles``.

Same functions, keyword arguments, dict schemas and exceptions as the reference (speckles.py:83-817).  The
array work runs in libb4d: autocorrelation / PSD (one forward transform each), moments, percentile selection,
polar radial profile, PSD disc statistics, phase-correlation tracking; only the 1-D crossing searches on a few
thousand samples stay on the host.  Differences: device arithmetic is float32; FFT-based
groups need power-of-two frame sizes, so FFT-based TILE statistics (170/171- or 227/228-pixel tiles) are
skipped with a RuntimeWarning until general-length plans exist; the default tracker of the reference
(`tracking_method="template"`, cv2 / scikit-image NCC) runs on the device through ``b4d_template_match`` (parity with
those libraries is unpinned: they are absent from the build image, the oracle follows their published definition).
"""
from __future__ import annotations

import logging
import threading
import warnings
from typing import Literal, Sequence

import numpy as np

from .. import _device as D
from .. import _ffi
from ..geometry.roi import odd_size, roi_grid_3x3
from ..maths.radial import radial_mean_binned, radial_profile_batch
from ..maths.stats import (distance_at_fraction_from_peak, distances_at_fraction_from_peak_batch, width_at_fraction,
                           widths_at_fraction_batch)
from ..signal import corr as _corr
from ..signal import fft as _fft
from . import kernels as K
from .common import (choose_tiling_mode, grids_to_fields, normalize_groups, stack_time_series,
                     tile_spans, tiles_meta)
from .statistics import moments_from_sums

logger = logging.getLogger(__name__)

_SPECKLE_UNITS: dict[str, dict[str, str]] = {
    "amplitude": {"visibility": "", "contrast": ""},
    "stats": {"mean": "a.u.", "std": "a.u.", "variance": "a.u.^2", "skewness": "", "kurtosis": "", "frac_zero": "",
              "frac_sat": "", "SNRdB": "dB"},
    "grain": {"lx": "px", "ly": "px", "leq": "px", "r": "", "xlag": "px", "ylag": "px", "autocorr": ""},
    "bandwidth": {"spr": "", "feq": "1/px", "f95": "1/px", "sig_fx": "1/px", "sig_fy": "1/px", "rf": ""},
    "temporal": {"dx": "px", "dy": "px", "r": "px", "std_dx": "px", "std_dy": "px", "std_r": "px"},
}
_ALL_SPECKLE_GROUPS: set[str] = {"amplitude", "grain", "bandwidth", "stats"}
_FFT_GROUPS = {"grain", "bandwidth"}


# ------------------------------------------------------------------------------------------------ device helpers
def _dev2d(image):
    t, _, _ = D.to_device_f32(image, ndim=(2,))
    return t


def _pad_square_dev(t):
    """pad_to_square(fill = mean) on the device (geometry/masks.py:11-57)."""
    import torch

    h, w = t.shape
    if h == w:
        return t
    n = max(h, w)
    out = torch.full((n, n), float(K.moments_batch(_pad4(t[None]), eps=0.0, saturation=None)[0, 1]), dtype=torch.float32, device=t.device)
    y0, x0 = (n - h) // 2, (n - w) // 2
    out[y0:y0 + h, x0:x0 + w] = t
    return out


def _fft_ok(shape) -> bool:
    n = max(shape)
    return _ffi.supported(n, n) and _ffi.supported(int(shape[0]), int(shape[1]))


def _amplitude_from(mom_row, pct_row) -> dict:
    n, mean, m2 = float(mom_row[0]), float(mom_row[1]), float(mom_row[2])
    if not np.isfinite(mean) or mean <= 0.0 or n <= 0:
        raise ValueError("Mean intensity must be positive and finite.")
    vmin, vmax = float(pct_row[0]), float(pct_row[1])
    denom = vmax + vmin
    if not np.isfinite(denom) or denom <= 0.0:
        raise ValueError("Invalid percentile range for Michelson contrast.")
    return {"visibility": float(np.sqrt(m2 / n)) / mean, "contrast": (vmax - vmin) / denom}


def _widths_from_autocorr(ac_dev, fraction: float, radial_method: str):
    """lx, ly (1/e full widths of the cuts through the peak) and leq (from the radial mean) of ONE device map."""
    n_y, n_x = (int(v) for v in ac_dev.shape)
    flat = int(ac_dev.argmax())
    iy, ix = divmod(flat, n_x)
    y_cut = ac_dev[:, ix].double().cpu().numpy()
    x_cut = ac_dev[iy, :].double().cpu().numpy()
    ly, _ = width_at_fraction(y_cut, fraction=fraction, center_index=iy)
    lx, _ = width_at_fraction(x_cut, fraction=fraction, center_index=ix)
    if radial_method == "binned":
        rad, r = radial_mean_binned(ac_dev)
    else:
        rad2, r = radial_profile_batch(ac_dev[None])
        rad = rad2[0]
    if rad.size < 2 or r.size < 2:
        raise ValueError("Radial profile is too short to estimate leq.")
    dr = float(r[1] - r[0])
    if dr <= 0:
        raise ValueError("Invalid radial sampling (non-positive dr).")
    dist, _ = distance_at_fraction_from_peak(rad, fraction=fraction, peak_index=0)
    return float(lx), float(ly), 2 * float(dist) * dr


def _bandwidth_from(stats_row) -> dict:
    s, sfr, sfx, sfy, sp2, _, _, f95 = (float(v) for v in stats_row)
    if not np.isfinite(s) or s <= 0.0:
        raise ValueError("PSD energy is not positive/finite after mean/DC removal.")
    sig_fx, sig_fy = float(np.sqrt(sfx / s)), float(np.sqrt(sfy / s))
    den = sp2 / (s * s)
    if not np.isfinite(den) or den <= 0.0:
        raise ValueError("Invalid SPR denominator (unexpected).")
    return {"feq": float(np.sqrt(sfr / s)), "f95": f95, "sig_fx": sig_fx, "sig_fy": sig_fy,
            "rf": float(sig_fx / sig_fy) if sig_fy != 0.0 else float("inf"), "spr": float(1.0 / den)}


def _pad_square_batch(stack):
    """pad_to_square(fill = mean of each item) for a (B, h, w) device stack (geometry/masks.py:11-57)."""
    b, h, w = (int(v) for v in stack.shape)
    if h == w:
        return stack
    n = max(h, w)
    means = K.moments_batch(_pad4(stack), eps=0.0, saturation=None)[:, 1].float()     # float64 means from b4d_moments
    out = means[:, None, None].expand(b, n, n).contiguous()
    y0, x0 = (n - h) // 2, (n - w) // 2
    out[:, y0:y0 + h, x0:x0 + w] = stack
    return out


def _widths_batch(ac, fraction: float):
    """[(lx, ly, leq)] of a (B, n, n) device stack of autocorrelation maps (interpolated radial mean): one arg-max, one
    gather of the two cuts and one radial-profile launch for the whole batch, then the 1-D crossing searches on the host."""
    import torch

    b, n_y, n_x = (int(v) for v in ac.shape)
    flat = ac.reshape(b, -1).argmax(dim=1)
    iy, ix = flat // n_x, flat % n_x
    ar = torch.arange(b, device=ac.device)
    y_cuts = ac[ar, :, ix].double().cpu().numpy()
    x_cuts = ac[ar, iy, :].double().cpu().numpy()
    iy, ix = iy.cpu().numpy(), ix.cpu().numpy()
    rad, r = radial_profile_batch(ac)
    if rad.shape[1] < 2 or r.size < 2:
        raise ValueError("Radial profile is too short to estimate leq.")
    dr = float(r[1] - r[0])
    if dr <= 0:
        raise ValueError("Invalid radial sampling (non-positive dr).")
    ly, _ = widths_at_fraction_batch(y_cuts, iy, fraction=fraction)
    lx, _ = widths_at_fraction_batch(x_cuts, ix, fraction=fraction)
    dist, _ = distances_at_fraction_from_peak_batch(rad, fraction=fraction, peak_index=0)
    return [(float(lx[i]), float(ly[i]), 2 * float(dist[i]) * dr) for i in range(b)]


def _grain_batch(stack, fraction: float = 1.0 / np.e) -> list[dict]:
    """lx, ly, leq, r of every item of a (B, h, w) device stack (the tile path of `grain`)."""
    if min(int(stack.shape[1]), int(stack.shape[2])) < 128:
        raise ValueError("image too small for speckle grain metrics (min dimension < 128).")
    ac = _corr.autocorr2d_stack(_pad_square_batch(stack), remove_mean=True, standardize=False, normalize="peak", return_tensors=True)
    return [{"lx": lx, "ly": ly, "leq": float(leq), "r": float(lx / ly) if ly != 0 else float("inf")}
            for lx, ly, leq in _widths_batch(ac, fraction)]


def _bandwidth_batch(stack) -> list[dict]:
    """bandwidth() of every item of a (B, h, w) device stack."""
    sq = _pad_square_batch(stack)
    if not bool(sq.isfinite().all()):
        raise ValueError("image mean is not finite.")
    psd = _fft.psd2d_stack(sq, scale=True, return_tensors=True)
    return [_bandwidth_from(row) for row in K.psd_stats_batch(psd)]


def tiled_fields_batched_multi(tb, tile_mode: str, batch_fn) -> list[dict]:
    """Per frame of a (B, H, W) device stack: {key: {"mean": (3, 3), "std": (3, 3)}} from batch_fn((k, th, tw) stack) ->
    [dict], evaluated once per tile shape over the tiles of ALL frames (the batched form of metrics/common.py:278-378)."""
    b = int(tb.shape[0])
    n, batches = _tile_batches(tb, tile_mode)
    grids = None
    for _, frcs, stack in batches:
        vals = batch_fn(stack)
        if grids is None:
            grids = {k: np.empty((b, n, n), dtype=float) for k in vals[0]}
        for (f, r, c), v in zip(frcs, vals):
            for k in grids:
                grids[k][f, r, c] = float(v[k])
    return [grids_to_fields({k: g[f] for k, g in grids.items()}, n) for f in range(b)]


def tiled_fields_batched(t, tile_mode: str, batch_fn) -> dict:
    """Single-frame form of tiled_fields_batched_multi."""
    return tiled_fields_batched_multi(t[None], tile_mode, batch_fn)[0]


# ------------------------------------------------------------------------------------------------ metric functions
def grain(image, *, fraction: float = 1.0 / np.e, radial_method: Literal["binned", "interpolated"] = "interpolated",
          verbose: bool = False) -> dict:
    """Speckle grain size from the autocorrelation peak: lx, ly, leq, r = lx/ly, autocorr, xlag, ylag
    (reference: speckles.py:497-596)."""
    data = image if D.is_tensor(image) else np.asarray(image)
    if data.ndim != 2:
        raise ValueError("image must be a 2D array.")
    if min(data.shape) < 128:
        raise ValueError("image too small for speckle grain metrics (min dimension < 128).")
    if radial_method not in ("binned", "interpolated"):
        raise ValueError("radial_method must be 'binned' or 'interpolated'.")
    sq = _pad_square_dev(_dev2d(data))
    ac, xlag, ylag = _corr.autocorr2d(sq, dx=1.0, dy=1.0, remove_mean=True, standardize=False, normalize="peak",
                                      return_tensors=True)
    lx, ly, leq = _widths_from_autocorr(ac, fraction, radial_method)
    out = {"lx": lx, "ly": ly, "leq": float(leq), "r": float(lx / ly) if ly != 0 else float("inf"),
           "autocorr": D.to_host(ac, np.float64), "xlag": np.asarray(xlag, dtype=float), "ylag": np.asarray(ylag, dtype=float)}
    if verbose:
        logger.info("> grain: lx=%.2f | ly=%.2f | lx/ly=%.2f | leq=%.2f ", out["lx"], out["ly"], out["r"], out["leq"])
    return out


def amplitude(image, verbose: bool = False) -> dict:
    """visibility = std/mean and robust Michelson contrast from the 0.05 / 99.95 percentiles
    (reference: speckles.py:602-663)."""
    data = image if D.is_tensor(image) else np.asarray(image)
    if data.ndim != 2:
        raise ValueError("image must be a 2D array.")
    t = _dev2d(data)[None]
    mom = K.moments_batch(_pad4(t), eps=0.0, saturation=None).cpu().numpy()[0]
    pct = K.percentiles_batch(t, [0.05, 99.95])[0]
    out = _amplitude_from(mom, pct)
    if verbose:
        logger.info("> visibility: %.2f | contrast: %.2f", out["visibility"], out["contrast"])
    return out


def _pad4(t):
    """(B, ...) -> (B, npix padded to a multiple of 4 with NaN) for the 16-byte-vector moment kernel."""
    import torch

    flat = t.reshape(t.shape[0], -1)
    pad = (-flat.shape[1]) % 4
    if pad:
        flat = torch.cat([flat, torch.full((flat.shape[0], pad), float("nan"), device=flat.device)], dim=1)
    return flat


def bandwidth(image, verbose: bool = False) -> dict[str, float]:
    """Spatial-frequency bandwidth metrics of the PSD over the inscribed frequency disc: feq, f95, sig_fx, sig_fy,
    rf, spr (reference: speckles.py:669-817)."""
    data = image if D.is_tensor(image) else np.asarray(image)
    if data.ndim != 2:
        raise ValueError("image must be a 2D array.")
    sq = _pad_square_dev(_dev2d(data))
    if not bool(sq.isfinite().all()):
        raise ValueError("image mean is not finite.")
    psd = _fft.psd2d_stack(sq[None], scale=True, return_tensors=True)   # mean removal == zeroed DC bin (kernel)
    out = _bandwidth_from(K.psd_stats_batch(psd)[0])
    if verbose:
        logger.info("> bandwidth: fx=%.4f | fy=%.4f | fx/fy=%.2f | feq=%.4f | f95=%.4f | spr=%.0f", out["sig_fx"],
                    out["sig_fy"], out["rf"], out["feq"], out["f95"], out["spr"])
    return out


# ------------------------------------------------------------------------------------------------ tiles
class tile_batch_memo:
    """Within one aggregator call the tile batches of the frame are built once and shared by the metric groups (81 slices and
    four concatenations per 2048-px frame: ~1 ms of host time each time); keyed by buffer, shape and tile mode, dropped at exit."""
    _local = threading.local()

    def __enter__(self):
        self._prev = getattr(self._local, "memo", None)
        self._local.memo = {} if self._prev is None else self._prev
        return self

    def __exit__(self, *exc):
        self._local.memo = self._prev
        return False

    @classmethod
    def current(cls):
        return getattr(cls._local, "memo", None)

    @classmethod
    def scoped(cls, fn):
        """Decorator: the wrapped aggregator runs inside one memo."""
        import functools

        @functools.wraps(fn)
        def run(*a, **kw):
            with cls():
                return fn(*a, **kw)
        return run


def _tile_batches(tb, tile_mode: str):
    """Tiles of a (B, H, W) device stack grouped by shape -> (n, [(shape, [(frame, r, c), ...], (k, th, tw) tensor)]);
    within a group the tiles are ordered tile-major, frame-minor."""
    memo = tile_batch_memo.current()
    if memo is not None:
        key = (int(tb.data_ptr()), tuple(int(v) for v in tb.shape), tuple(int(v) for v in tb.stride()), tile_mode)
        hit = memo.get(key)
        if hit is None:
            hit = memo[key] = _tile_batches_build(tb, tile_mode)
        return hit
    return _tile_batches_build(tb, tile_mode)


def _tile_batches_build(tb, tile_mode: str):
    import torch

    b = int(tb.shape[0])
    n, ys, xs = tile_spans(int(tb.shape[1]), int(tb.shape[2]), tile_mode)
    groups: dict[tuple[int, int], list] = {}
    for r, (y0, y1) in enumerate(ys):
        for c, (x0, x1) in enumerate(xs):
            groups.setdefault((y1 - y0, x1 - x0), []).append((r, c, tb[:, y0:y1, x0:x1]))
    out = []
    for shape, items in groups.items():
        frcs = [(f, r, c) for r, c, _ in items for f in range(b)]
        out.append((shape, frcs, torch.cat([v for _, _, v in items], dim=0).contiguous()))
    return n, out


def _tiles_pointwise_multi(tb, tile_mode: str, want_amp: bool, want_stats: bool, saturation_value, eps) -> list[dict]:
    """Per frame: amplitude / stats tile grids from batched kernels over the tiles of all frames."""
    b = int(tb.shape[0])
    n, batches = _tile_batches(tb, tile_mode)
    amp = {k: np.empty((b, n, n)) for k in ("visibility", "contrast")} if want_amp else None
    st = None
    for _, frcs, stack in batches:
        mom = K.moments_batch(_pad4(stack), eps=eps, saturation=saturation_value).cpu().numpy()
        pct = K.percentiles_batch(stack, [0.05, 99.95]) if want_amp else None
        for i, (f, r, c) in enumerate(frcs):
            if want_amp:
                a = _amplitude_from(mom[i], pct[i])
                for k in amp:
                    amp[k][f, r, c] = a[k]
            if want_stats:
                d = moments_from_sums(mom[i], saturation_value)
                if st is None:
                    st = {k: np.empty((b, n, n)) for k in d}
                for k in st:
                    st[k][f, r, c] = d[k]
    outs = []
    for f in range(b):
        out = {}
        if want_amp:
            out["amplitude"] = grids_to_fields({k: g[f] for k, g in amp.items()}, n)
        if want_stats:
            out["stats"] = grids_to_fields({k: g[f] for k, g in st.items()}, n)
        outs.append(out)
    return outs


def _tiles_pointwise(t, tile_mode: str, want_amp: bool, want_stats: bool, saturation_value, eps):
    return _tiles_pointwise_multi(t[None], tile_mode, want_amp, want_stats, saturation_value, eps)[0]


@tile_batch_memo.scoped
def speckle_stats_batch(tb, *, groups: set, tiles: bool = True, saturation_value: float | None = 65535.0, eps: float = 1e-6,
                        keep_autocorr: bool = True, autocorr_out: np.ndarray | None = None) -> list[dict]:
    """{"full": ..., "tiles": ...} of every frame of a (B, H, W) device stack (already in display orientation): the
    arithmetic of speckle_stats with every kernel launched once per batch and tile shape instead of once per frame and
    tile.  Used by speckle_stack_stats; equal to the per-frame function (tests/test_gpu_metrics.py)."""
    b, h, w = (int(v) for v in tb.shape)
    outs = [{"full": {}} for _ in range(b)]
    if groups & {"amplitude", "stats"}:
        mom = K.moments_batch(_pad4(tb), eps=eps, saturation=saturation_value).cpu().numpy()
        pct = K.percentiles_batch(tb, [0.05, 99.95]) if "amplitude" in groups else None
        for f in range(b):
            if "amplitude" in groups:
                outs[f]["full"]["amplitude"] = _amplitude_from(mom[f], pct[f])
    if "grain" in groups:
        if min(h, w) < 128:
            raise ValueError("image too small for speckle grain metrics (min dimension < 128).")
        sq = _pad_square_batch(tb)
        ac = _corr.autocorr2d_stack(sq, remove_mean=True, standardize=False, normalize="peak", return_tensors=True)
        n_sq = int(sq.shape[1])
        xlag = (np.arange(n_sq) - n_sq // 2).astype(float)
        if autocorr_out is not None:     # (B, n, n) float64 slice of the caller's (T, n, n) result: converted on the device, one copy
            import torch

            torch.from_numpy(autocorr_out).copy_(ac.double())
            ac_host = autocorr_out
        else:
            ac_host = D.to_host(ac, np.float64) if keep_autocorr else None
        for f, (lx, ly, leq) in enumerate(_widths_batch(ac, 1.0 / np.e)):
            g = {"lx": lx, "ly": ly, "leq": float(leq), "r": float(lx / ly) if ly != 0 else float("inf")}
            if ac_host is not None:
                g.update({"autocorr": ac_host[f], "xlag": xlag, "ylag": xlag.copy()})
            outs[f]["full"]["grain"] = g
    if "stats" in groups:
        for f in range(b):
            outs[f]["full"]["stats"] = moments_from_sums(mom[f], saturation_value)
    if "bandwidth" in groups:
        for f, d in enumerate(_bandwidth_batch(tb)):
            outs[f]["full"]["bandwidth"] = d
    mode, tile_shape_px = choose_tiling_mode(h, w, tiles=tiles, min_tile_px=128)
    if mode == "off":
        return outs
    tiles_out = _tiles_pointwise_multi(tb, mode, "amplitude" in groups, "stats" in groups, saturation_value, eps)
    fft_groups = sorted(groups & _FFT_GROUPS)
    if fft_groups:
        n, ys, xs = tile_spans(h, w, mode)
        if all(_fft_ok((y1 - y0, x1 - x0)) for y0, y1 in ys for x0, x1 in xs):
            if "grain" in groups:
                for f, d in enumerate(tiled_fields_batched_multi(tb, mode, _grain_batch)):
                    tiles_out[f]["grain"] = d
            if "bandwidth" in groups:
                for f, d in enumerate(tiled_fields_batched_multi(tb, mode, _bandwidth_batch)):
                    tiles_out[f]["bandwidth"] = d
        else:
            warnings.warn(f"tile statistics of {fft_groups} skipped: {tile_shape_px}-pixel tiles have no transform plan; "
                          "full-frame values are unaffected.", RuntimeWarning, stacklevel=2)
    for f in range(b):
        if tiles_out[f]:
            outs[f]["tiles"] = {g: tiles_out[f][g] for g in ("amplitude", "grain", "stats", "bandwidth") if g in tiles_out[f]}
    return outs


# ------------------------------------------------------------------------------------------------ aggregators
@tile_batch_memo.scoped
def speckle_stats(image: np.ndarray, *, metrics: str | Sequence[str] = "all", tiles: bool = True,
                  display_origin: Literal["upper", "lower"] = "lower", saturation_value: float | None = 65535.0,
                  eps: float = 1e-6, verbose: bool = True) -> dict:
    """Speckle metrics of one 2-D image: {"meta", "full", "tiles"} (reference: speckles.py:83-255)."""
    if not isinstance(image, np.ndarray):
        raise TypeError("speckle_stats expects a numpy.ndarray")
    if image.ndim != 2:
        raise ValueError(f"Expected 2D array, got ndim={image.ndim}")
    from .common import normalize_display_origin

    lower = normalize_display_origin(display_origin) == "lower"   # apply_display_origin (common.py:44-72): rows flipped before
    h, w = image.shape                                            # the metrics -- on the device (a strided host copy costs 1 ms)
    groups = normalize_groups(metrics, all_groups=_ALL_SPECKLE_GROUPS, context="speckles", param_name="metrics")
    if verbose:
        logger.info("\nspeckle stats for a (h x w: %.0f x %.0f) image:", h, w)
    out: dict = {"meta": {"kind": "speckles", "display_origin": display_origin, "input_shape": (int(h), int(w)),
                          "requested_groups": sorted(groups), "units": _SPECKLE_UNITS}, "full": {}}
    t = _dev2d(np.ascontiguousarray(image))
    if lower:
        t = t.flip(0).contiguous()
    if "amplitude" in groups:
        out["full"]["amplitude"] = amplitude(t, verbose=verbose)
    if "grain" in groups:
        out["full"]["grain"] = grain(t, verbose=verbose)
    if "stats" in groups:
        from .statistics import distribution_moments

        out["full"]["stats"] = distribution_moments(t, saturation_value=saturation_value, eps=eps, verbose=verbose)
    if "bandwidth" in groups:
        out["full"]["bandwidth"] = bandwidth(t, verbose=verbose)

    mode, tile_shape_px = choose_tiling_mode(h, w, tiles=tiles, min_tile_px=128)
    if mode == "off":
        return out
    out["meta"].update(tiles_meta(h, w, tile_mode=mode, tile_shape_px=tile_shape_px))
    tiles_out = _tiles_pointwise(t, mode, "amplitude" in groups, "stats" in groups, saturation_value, eps)
    fft_groups = sorted(groups & _FFT_GROUPS)
    if fft_groups:
        n, ys, xs = tile_spans(h, w, mode)
        native = all(_fft_ok((y1 - y0, x1 - x0)) for y0, y1 in ys for x0, x1 in xs)
        if native:
            if "grain" in groups:      # all tiles of one shape per launch
                tiles_out["grain"] = tiled_fields_batched(t, mode, _grain_batch)
            if "bandwidth" in groups:
                tiles_out["bandwidth"] = tiled_fields_batched(t, mode, _bandwidth_batch)
        else:
            warnings.warn(f"tile statistics of {fft_groups} skipped: {tile_shape_px}-pixel tiles need a general-length "
                          "FFT plan (not built yet); full-frame values are unaffected.", RuntimeWarning, stacklevel=2)
    if tiles_out:
        out["tiles"] = {g: tiles_out[g] for g in ("amplitude", "grain", "stats", "bandwidth") if g in tiles_out}
    return out


def speckle_stack_stats(stack: np.ndarray, *, metrics: str | Sequence[str] = "all", tiles: bool = True,
                        display_origin: Literal["upper", "lower"] = "lower", roi_grain_factor: float = 3.0,
                        roi_step_factor: float = 0.5, tracking_method: str = "template",
                        tracking_backend: Literal["internal", "skimage", "opencv"] = "skimage", subpixel: bool = True,
                        saturation_value: float | None = 65535.0, eps: float = 1e-6, verbose: bool = True,
                        parallel: bool = True, n_jobs: int | None = None) -> dict:
    """Per-frame speckle metrics stacked along T plus abs / inc translation tracking on a central 3x3 ROI grid
    (reference: speckles.py:258-490).  `parallel` / `n_jobs` are accepted for signature compatibility; frames
    are batched on the device instead of joblib threads."""
    if not isinstance(stack, np.ndarray):
        raise TypeError("speckle_stack_stats expects a numpy.ndarray")
    if stack.ndim != 3:
        raise ValueError(f"stack must be a 3D array with shape (T, H, W); got ndim={stack.ndim}")
    T, H, W = (int(v) for v in stack.shape)
    if T < 1:
        raise ValueError("stack must contain at least one frame.")
    method = tracking_method.strip().lower()
    if method == "template":
        if tracking_backend not in ("opencv", "skimage"):
            raise ValueError("backend must be 'opencv' or 'skimage'.")
    elif method == "phase":
        if tracking_backend not in ("internal", "skimage"):
            raise ValueError("backend must be 'internal' or 'skimage'.")
    else:
        raise ValueError(f"Unsupported tracking method: {tracking_method!r}. Supported: phase, template")

    groups = normalize_groups(metrics, all_groups=_ALL_SPECKLE_GROUPS, context="speckles", param_name="metrics")
    from .common import normalize_display_origin

    lower = normalize_display_origin(display_origin) == "lower"
    dev_all, _, _ = D.to_device_f32(stack, ndim=(3,))
    fb = max(1, min(T, (256 << 20) // (4 * H * W)))     # frames per batch: <= 256 MiB of pixels
    per_frame = []
    n_sq = max(H, W)
    ac_all = np.empty((T, n_sq, n_sq), dtype=np.float64) if "grain" in groups else None   # full.grain.autocorr, as the reference stacks it
    for a in range(0, T, fb):
        tb = dev_all[a:a + fb]
        if lower:                                       # apply_display_origin (common.py:44-72): rows flipped before the metrics
            tb = tb.flip(1).contiguous()
        per_frame.extend(speckle_stats_batch(tb, groups=groups, tiles=tiles, saturation_value=saturation_value, eps=eps,
                                             autocorr_out=None if ac_all is None else ac_all[a:a + fb]))
    if ac_all is not None:
        for d in per_frame:
            d["full"]["grain"].pop("autocorr")
    out_full = stack_time_series([d["full"] for d in per_frame])
    if ac_all is not None:
        out_full["grain"]["autocorr"] = ac_all
    out_tiles = None
    if tiles and all(isinstance(d.get("tiles"), dict) for d in per_frame):
        out_tiles = stack_time_series([d["tiles"] for d in per_frame])

    grain0 = grain(stack[0], verbose=False)
    ell = float(np.nanmax([grain0.get("lx", np.nan), grain0.get("ly", np.nan), grain0.get("leq", np.nan)]))
    if not np.isfinite(ell) or ell <= 0:
        raise ValueError("Could not infer a valid grain size from frame 0 (lx/ly/leq).")
    roi_side = odd_size(int(np.ceil(roi_grain_factor * ell)))
    step = int(max(1, round(roi_step_factor * roi_side)))
    grid_slices, grid_labels = roi_grid_3x3((H, W), (roi_side, roi_side), (step, step), center_yx=None)
    rois = [(s[0].start, s[0].stop, s[1].start, s[1].stop) for s in grid_slices.ravel()]

    # abs: frame-0 templates (9, shared by every t); inc: templates cut from frame t-1 (frame 0 for t = 0).  Frames go
    # through the tracker in blocks (metrics/sharded.py: track_abs_inc) so that the resident spectra stay bounded.
    from .sharded import track_abs_inc

    res_abs, res_inc = track_abs_inc(dev_all, dev_all[0], dev_all[0], rois, method=method,
                                     backend=tracking_backend, subpixel=subpixel, eps=1e-9)
    dy_abs = res_abs[:, 0].reshape(T, 3, 3).astype(np.float32)
    dx_abs = res_abs[:, 1].reshape(T, 3, 3).astype(np.float32)
    dy_inc = res_inc[:, 0].reshape(T, 3, 3).astype(np.float32)
    dx_inc = res_inc[:, 1].reshape(T, 3, 3).astype(np.float32)

    def block(dx, dy):
        r = np.sqrt(dx ** 2 + dy ** 2)
        red = lambda fn, a: fn(a, axis=(1, 2)).astype(np.float32)  # noqa: E731
        return {"dx": red(np.nanmean, dx), "dy": red(np.nanmean, dy), "r": red(np.nanmean, r),
                "std_dx": red(np.nanstd, dx), "std_dy": red(np.nanstd, dy), "std_r": red(np.nanstd, r)}

    meta = {"kind": "speckle_stack_stats", "input_shape": (H, W), "stack_shape": (T, H, W), "n_frames": T,
            "display_origin": display_origin, "units": _SPECKLE_UNITS,
            "grain0": {k: grain0.get(k) for k in ("lx", "ly", "leq", "r")},
            "tracking": {"method": str(tracking_method), "backend": str(tracking_backend), "subpixel": bool(subpixel),
                         "peak_mode": "abs", "search_area": "full_frame",
                         "normalization": {"template": "zscore_local", "search": "zscore_global"},
                         "roi_grain_factor": float(roi_grain_factor), "roi_size_yx": (int(roi_side), int(roi_side)),
                         "roi_step_factor": float(roi_step_factor), "roi_step_yx": (int(step), int(step)),
                         "roi_labels": grid_labels, "roi_order": "row-major"},
            "parallel": {"enabled": False, "joblib_verbose": 0}}
    out = {"meta": meta, "full": out_full,
           "temporal": {"abs": block(dx_abs, dy_abs), "inc": block(dx_inc, dy_inc), "qc": {"roi_grid_shape": (3, 3)}}}
    if out_tiles is not None:
        out["tiles"] = out_tiles
    if verbose:
        logger.info("> speckle_stack_stats | frames=%d | roi=%dx%d | step=%d | device batch", T, roi_side, roi_side, step)
    return out
