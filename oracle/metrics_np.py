"""Oracle (test infrastructure, NOT product code): NumPy/SciPy restatement of
``barc4dip.metrics`` (speckles, sharpness, statistics, common) and the
``barc4dip.maths`` helpers they call.

Citations are ``file:line`` under /root/reference/src/barc4dip.  Parity pinned
by tests/golden/metrics_*.npz (made by oracle/make_golden.py from the reference).
"""
from __future__ import annotations

import warnings

import numpy as np
from scipy import ndimage
from scipy.interpolate import RegularGridInterpolator
from scipy.stats import describe

from .signal_np import (autocorr2d, odd_size, pad_to_square, psd2d, roi_grid_3x3,
                        track_translation)

INV_E = 1.0 / np.e


# --------------------------------------------------------------------------- maths
def width_at_fraction(profile, *, fraction=INV_E, center_index=None):
    """maths/stats.py:9-89 -- full width at fraction*peak, linear-interp crossings;
    (size, True) when a side never drops below the threshold."""
    p = np.asarray(profile, dtype=float)
    if p.ndim != 1 or p.size == 0:
        raise ValueError("profile must be a non-empty 1D array.")
    if not (0.0 < fraction < 1.0):
        raise ValueError("fraction must be in (0, 1).")
    c = int(np.argmax(p) if center_index is None else center_index)
    c = max(0, min(c, p.size - 1))
    thr = p[c] * fraction
    below = p < thr
    left = np.flatnonzero(below[: c + 1])
    right = np.flatnonzero(below[c:])
    if left.size == 0 or right.size == 0:
        return float(p.size), True
    i0 = int(left[-1])
    j1 = c + int(right[0])
    ya, yb = p[i0], p[i0 + 1]
    xl = float(i0) if yb == ya else i0 + (thr - ya) / (yb - ya)
    ya, yb = p[j1 - 1], p[j1]
    xr = float(j1) if yb == ya else (j1 - 1) + (thr - ya) / (yb - ya)
    return float(xr - xl), False


def distance_at_fraction_from_peak(profile, *, fraction=INV_E, peak_index=0):
    """maths/stats.py:92-156 -- one-sided distance to the first sample below fraction*peak."""
    p = np.asarray(profile, dtype=float)
    if p.ndim != 1 or p.size == 0:
        raise ValueError("profile must be a non-empty 1D array.")
    if not (0.0 < fraction < 1.0):
        raise ValueError("fraction must be in (0, 1).")
    k0 = max(0, min(int(peak_index), p.size - 1))
    thr = p[k0] * fraction
    hit = np.flatnonzero(p[k0:] < thr)
    if hit.size == 0:
        return float(p.size), True
    i1 = k0 + int(hit[0])
    if i1 == k0:
        return 0.0, False
    ya, yb = p[i1 - 1], p[i1]
    xc = float(i1) if yb == ya else (i1 - 1) + (thr - ya) / (yb - ya)
    return float(xc - k0), False


def _pixel_axes(shape):
    ny, nx = shape
    return np.arange(nx, dtype=float) - (nx // 2), np.arange(ny, dtype=float) - (ny // 2)


def radial_mean_binned(signal_2d, *, r_max=None, bin_size=1.0):
    """maths/radial.py:38-98."""
    z = np.asarray(signal_2d, dtype=float)
    if z.ndim != 2:
        raise ValueError("signal_2d must be a 2D array.")
    if not np.isfinite(z).all():
        raise ValueError("signal_2d contains non-finite values.")
    if bin_size <= 0:
        raise ValueError("bin_size must be > 0.")
    x, y = _pixel_axes(z.shape)
    if r_max is None:
        r_max = min(float(np.max(np.abs(x))), float(np.max(np.abs(y))))
    if r_max <= 0:
        raise ValueError("r_max must be > 0.")
    R = np.sqrt(x[None, :] ** 2 + y[:, None] ** 2)
    nb = int(np.floor(r_max / bin_size)) + 1
    idx = np.floor(R / bin_size).astype(np.int64)
    m = idx < nb
    sums = np.bincount(idx[m].ravel(), weights=z[m].ravel(), minlength=nb).astype(float)
    cnt = np.bincount(idx[m].ravel(), minlength=nb).astype(float)
    out = np.full(nb, np.nan)
    out[cnt > 0] = sums[cnt > 0] / cnt[cnt > 0]
    return out, (np.arange(nb, dtype=float) + 0.5) * float(bin_size)


def radial_mean_interpolated(signal_2d, *, r_max=None, nr=None, ntheta=None, fill_value=0.0):
    """maths/radial.py:101-169 -- nr x int(2*pi*180)=1130 polar samples, bilinear
    RegularGridInterpolator with fill_value outside, mean over theta."""
    z = np.asarray(signal_2d, dtype=float)
    if z.ndim != 2:
        raise ValueError("signal_2d must be a 2D array.")
    if not np.isfinite(z).all():
        raise ValueError("signal_2d contains non-finite values.")
    x, y = _pixel_axes(z.shape)
    if r_max is None:
        r_max = min(float(np.max(np.abs(x))), float(np.max(np.abs(y))))
    if r_max <= 0:
        raise ValueError("r_max must be > 0.")
    nr = int(np.floor(r_max)) + 1 if nr is None else nr
    ntheta = int(2.0 * np.pi * 180.0) if ntheta is None else ntheta
    if nr <= 1:
        raise ValueError("nr must be > 1.")
    if ntheta <= 3:
        raise ValueError("ntheta must be > 3.")
    r = np.linspace(0.0, r_max, nr)
    th = np.linspace(0.0, 2.0 * np.pi, ntheta, endpoint=False)
    Rg, Tg = np.meshgrid(r, th, indexing="ij")
    f = RegularGridInterpolator((y, x), z, bounds_error=False, fill_value=fill_value)
    pts = np.column_stack([(Rg * np.sin(Tg)).ravel(), (Rg * np.cos(Tg)).ravel()])
    return np.mean(f(pts).reshape(Rg.shape), axis=1), r


def percentile_minmax_range(image, p_low=0.05, p_high=99.95):
    """utils/range.py:44-54."""
    a = np.asarray(image)
    return float(np.nanpercentile(a, p_low)), float(np.nanpercentile(a, p_high))


# --------------------------------------------------------------------------- statistics
def distribution_moments(image, *, saturation_value=65535.0, eps=1e-6, verbose=False):
    """metrics/statistics.py:17-125 -- finite-only float64 moments (biased skew / Fisher
    kurtosis from scipy.stats.describe), zero / saturation fractions, SNR in dB."""
    d = np.asarray(image)
    if d.ndim not in (1, 2):
        raise ValueError(f"Expected 1D or 2D array, got ndim={d.ndim}")
    if d.size == 0:
        raise ValueError("distribution_moments received an empty image.")
    v = np.asarray(d, dtype=np.float64).ravel()
    ok = np.isfinite(v)
    if not ok.any():
        raise ValueError("distribution_moments received no finite values.")
    v = v[ok]
    mean = float(np.mean(v))
    std = float(np.std(v, ddof=0))
    ds = describe(v, axis=None)
    if std == 0.0:
        snr = float("inf") if mean > 0.0 else float("nan")
    else:
        q = mean / std
        snr = float(20.0 * np.log10(q)) if q > 0.0 else (float("-inf") if q == 0.0 else float("nan"))
    return {
        "mean": mean, "std": std, "variance": float(std * std),
        "skewness": float(ds.skewness), "kurtosis": float(ds.kurtosis),
        "frac_zero": float(np.mean(np.abs(v) <= eps)),
        "frac_sat": float("nan") if saturation_value is None else float(np.mean(v >= float(saturation_value))),
        "SNRdB": snr,
    }


# --------------------------------------------------------------------------- speckle metrics
def amplitude(image, verbose=False):
    """metrics/speckles.py:602-663."""
    img = np.asarray(image, dtype=float)
    if img.ndim != 2:
        raise ValueError("image must be a 2D array.")
    mu = float(np.nanmean(img))
    if not np.isfinite(mu) or mu <= 0.0:
        raise ValueError("Mean intensity must be positive and finite.")
    vis = float(np.nanstd(img)) / mu
    lo, hi = percentile_minmax_range(img)
    if not np.isfinite(hi + lo) or hi + lo <= 0.0:
        raise ValueError("Invalid percentile range for Michelson contrast.")
    return {"visibility": vis, "contrast": (hi - lo) / (hi + lo)}


def _ac_widths(ac, fraction, dr_check_msg):
    """Shared tail of grain / inverse_autocorr_width: speckles.py:546-572, sharpness.py:699-722."""
    iy, ix = np.unravel_index(int(np.argmax(ac)), ac.shape)
    ly, _ = width_at_fraction(ac[:, ix], fraction=fraction, center_index=iy)
    lx, _ = width_at_fraction(ac[iy, :], fraction=fraction, center_index=ix)
    rad, r = radial_mean_interpolated(ac)
    rad = np.asarray(rad, dtype=float)
    r = np.asarray(r, dtype=float)
    if rad.size < 2 or r.size < 2:
        raise ValueError(dr_check_msg)
    dr = float(r[1] - r[0])
    if dr <= 0:
        raise ValueError("Invalid radial sampling (non-positive dr).")
    dist, _ = distance_at_fraction_from_peak(rad, fraction=fraction, peak_index=0)
    return float(lx), float(ly), 2.0 * float(dist) * dr


def grain(image, *, fraction=INV_E, radial_method="interpolated", verbose=False):
    """metrics/speckles.py:497-596."""
    data = np.asarray(image, dtype=float)
    if data.ndim != 2:
        raise ValueError("image must be a 2D array.")
    if min(data.shape) < 128:
        raise ValueError("image too small for speckle grain metrics (min dimension < 128).")
    data = pad_to_square(data, fill_value=np.mean(data))
    ac, xlag, ylag = autocorr2d(data, dx=1.0, dy=1.0, remove_mean=True, standardize=False,
                                normalize="peak")
    if radial_method == "binned":
        iy, ix = np.unravel_index(int(np.argmax(ac)), ac.shape)
        ly, _ = width_at_fraction(ac[:, ix], fraction=fraction, center_index=iy)
        lx, _ = width_at_fraction(ac[iy, :], fraction=fraction, center_index=ix)
        rad, r = radial_mean_binned(ac)
        dist, _ = distance_at_fraction_from_peak(rad, fraction=fraction, peak_index=0)
        leq = 2 * float(dist) * float(r[1] - r[0])
        lx, ly = float(lx), float(ly)
    elif radial_method == "interpolated":
        lx, ly, leq = _ac_widths(ac, fraction, "Radial profile is too short to estimate leq.")
    else:
        raise ValueError("radial_method must be 'binned' or 'interpolated'.")
    return {"lx": lx, "ly": ly, "leq": float(leq),
            "r": float(lx / ly) if ly != 0 else float("inf"),
            "autocorr": np.asarray(ac, dtype=float),
            "xlag": np.asarray(xlag, dtype=float), "ylag": np.asarray(ylag, dtype=float)}


def bandwidth(image, verbose=False):
    """metrics/speckles.py:669-817 -- PSD moments over the inscribed frequency disc."""
    img = np.asarray(image, dtype=float)
    if img.ndim != 2:
        raise ValueError("image must be a 2D array.")
    img = pad_to_square(img, fill_value=np.mean(img))
    mu = float(np.nanmean(img))
    if not np.isfinite(mu):
        raise ValueError("image mean is not finite.")
    P, fx, fy = psd2d(img - mu, dx=1.0, dy=1.0, scale=True)
    P = np.nan_to_num(np.asarray(P, dtype=float), nan=0.0, posinf=0.0, neginf=0.0).copy()
    ny, nx = P.shape
    P[ny // 2, nx // 2] = 0.0
    FX, FY = np.meshgrid(np.asarray(fx, dtype=float), np.asarray(fy, dtype=float), indexing="xy")
    FR = np.sqrt(FX * FX + FY * FY)
    m = FR <= min(float(np.max(np.abs(fx))), float(np.max(np.abs(fy))))
    Pm, FXm, FYm, FRm = P[m], FX[m], FY[m], FR[m]
    tot = float(np.sum(Pm))
    if not np.isfinite(tot) or tot <= 0.0:
        raise ValueError("PSD energy is not positive/finite after mean/DC removal.")
    feq = float(np.sqrt(np.sum(FRm * FRm * Pm) / tot))
    sfx = float(np.sqrt(np.sum(FXm * FXm * Pm) / tot))
    sfy = float(np.sqrt(np.sum(FYm * FYm * Pm) / tot))
    order = np.argsort(FRm)
    cdf = np.cumsum(Pm[order]) / tot
    k = min(int(np.searchsorted(cdf, 0.95, side="left")), FRm.size - 1)
    pn = Pm / tot
    den = float(np.sum(pn * pn))
    if not np.isfinite(den) or den <= 0.0:
        raise ValueError("Invalid SPR denominator (unexpected).")
    return {"feq": feq, "f95": float(FRm[order][k]), "sig_fx": sfx, "sig_fy": sfy,
            "rf": float(sfx / sfy) if sfy != 0.0 else float("inf"), "spr": float(1.0 / den)}


# --------------------------------------------------------------------------- sharpness metrics
def _checked2d(image, who, need_all_finite=False):
    d = np.asarray(image)
    if d.ndim != 2:
        raise ValueError(f"Expected 2D array, got ndim={d.ndim}")
    if d.size == 0:
        raise ValueError(f"{who} received an empty image.")
    fin = np.isfinite(d)
    if need_all_finite:
        if not fin.all():
            raise ValueError(f"{who} requires all values to be finite.")
    elif not fin.any():
        raise ValueError(f"{who} received image with no finite values.")
    return d, fin


def tenengrad(image, *, eps=1e-12, verbose=False):
    """metrics/sharpness.py:405-476 -- scipy Sobel (mode='reflect'), mean of squares."""
    d, fin = _checked2d(image, "tenengrad")
    v = np.asarray(d, dtype=float)
    gx = ndimage.sobel(v, axis=1, mode="reflect")
    gy = ndimage.sobel(v, axis=0, mode="reflect")
    ex = float(np.mean((gx * gx)[fin]))
    ey = float(np.mean((gy * gy)[fin]))
    return {"tenengrad": float(ex + ey), "ex": ex, "ey": ey, "re": float(ex / (ey + float(eps)))}


def laplacian_variance(image, *, verbose=False):
    """metrics/sharpness.py:482-530."""
    d, fin = _checked2d(image, "laplacian_variance")
    lap = ndimage.laplace(np.asarray(d, dtype=float), mode="reflect")
    return float(np.var(lap[fin], ddof=0))


def spectral_entropy(image, *, remove_mean=True, remove_dc=True, eps=1e-30, verbose=False):
    """metrics/sharpness.py:536-629.  The reference overwrites its pad_to_square result on
    the next line (590-591), so NO padding happens; reproduced."""
    d, _ = _checked2d(image, "spectral_entropy", need_all_finite=True)
    v = np.asarray(d, dtype=float)
    if remove_mean:
        v = v - float(np.mean(v))
    P, _, _ = psd2d(v, scale=False)
    P = np.asarray(P, dtype=float)
    if np.any(P < 0):
        raise ValueError("psd2d returned negative PSD values (unexpected).")
    if remove_dc:
        P = P.copy()
        P[P.shape[0] // 2, P.shape[1] // 2] = 0.0
    s = float(np.sum(P))
    if not np.isfinite(s) or s <= 0.0:
        raise ValueError("PSD sum is non-positive; cannot compute spectral entropy.")
    p = P.ravel() / s
    M = int(p.size - 1) if remove_dc else int(p.size)
    if M < 2:
        raise ValueError("Insufficient number of spectral bins to compute normalized entropy.")
    p = np.clip(p, float(eps), None)
    return float(float(-np.sum(p * np.log(p))) / np.log(float(M)))


def inverse_autocorr_width(image, *, fraction=INV_E, radial_method="interpolated",
                           min_size_px=32, verbose=False):
    """metrics/sharpness.py:635-746 ("binned" also runs the interpolated estimator, 704-707)."""
    data = np.asarray(image, dtype=float)
    if data.ndim != 2:
        raise ValueError("image must be a 2D array.")
    if data.size == 0:
        raise ValueError("inverse_autocorr_width received an empty image.")
    if min(data.shape) < int(min_size_px):
        raise ValueError("image too small for inverse autocorrelation width.")
    if radial_method not in ("binned", "interpolated"):
        raise ValueError("radial_method must be 'binned' or 'interpolated'.")
    data = pad_to_square(data, fill_value=np.mean(data))
    ac, _, _ = autocorr2d(data, dx=1.0, dy=1.0, remove_mean=True, standardize=True,
                          normalize="peak")
    lx, ly, leq = _ac_widths(ac, fraction, "Radial profile is too short to estimate equivalent width.")
    inv = lambda t: float(1.0 / t) if t != 0.0 else float("inf")  # noqa: E731
    return {"sx": inv(lx), "sy": inv(ly), "seq": inv(float(leq)),
            "r": float(lx / ly) if ly != 0.0 else float("inf")}


def eigenvalues(image, *, k=5, eps=1e-30, verbose=False):
    """metrics/sharpness.py:752-861 -- STA2: singular values of the energy-normalised,
    mean-removed image; eig = s^2/(M*N-1)."""
    d, _ = _checked2d(image, "eigenvalues", need_all_finite=True)
    if int(k) < 1:
        raise ValueError("k must be >= 1.")
    v = np.asarray(d, dtype=float)
    energy = float(np.sqrt(np.sum(v * v)))
    if not np.isfinite(energy) or energy <= 0.0:
        raise ValueError("eigenvalues cannot normalize an all-zero image.")
    J = v / energy
    J = J - float(np.mean(J))
    den = float(J.shape[0] * J.shape[1] - 1)
    if den <= 0.0:
        raise ValueError("eigenvalues requires at least 2 pixels (M*N >= 2).")
    s = np.linalg.svd(J, full_matrices=False, compute_uv=False)
    eig = (s * s) / den
    e1 = float(eig[0]) if eig.size >= 1 else 0.0
    e2 = float(eig[1]) if eig.size >= 2 else 0.0
    return {"eigenvalues": float(np.sum(eig[: min(int(k), int(eig.size))])), "e1": e1, "e2": e2,
            "re": float(e1 / (e2 + float(eps)))}


# --------------------------------------------------------------------------- tiling (metrics/common.py)
TILE_LABELS = np.array([["NW", "N", "NE"], ["W", "C", "E"], ["SW", "S", "SE"]], dtype=object)


def apply_display_origin(image, *, display_origin):
    """metrics/common.py:44-72 -- "lower" flips rows (a view) before any metric."""
    img = np.asarray(image)
    if img.ndim != 2:
        raise ValueError(f"apply_display_origin expects a 2D array, got ndim={img.ndim}")
    o = str(display_origin).strip().lower()
    if o not in ("upper", "lower"):
        raise ValueError("display_origin must be 'upper' or 'lower'.")
    return img[::-1, :] if o == "lower" else img


def split_edges(length, n_parts):
    """metrics/common.py:75-106 -- round(linspace) edges, last edge forced to length."""
    if length < 1:
        raise ValueError("length must be >= 1.")
    if n_parts < 1:
        raise ValueError("n_parts must be >= 1.")
    e = np.linspace(0, length, n_parts + 1)
    out = []
    for i in range(n_parts):
        a = int(round(float(e[i])))
        out.append((a, max(int(round(float(e[i + 1]))), a + 1)))
    out[-1] = (out[-1][0], length)
    return out


def choose_tiling_mode(h, w, *, tiles=False, min_tile_px=128):
    """metrics/common.py:109-170."""
    if h < 1 or w < 1:
        raise ValueError("Invalid image shape (h and w must be >= 1).")
    if min_tile_px < 1:
        raise ValueError("min_tile_px must be >= 1.")
    if not bool(tiles):
        return "off", None
    if h // 9 >= min_tile_px and w // 9 >= min_tile_px:
        return "subtiles_9x9", (h // 9, w // 9)
    if h // 3 >= min_tile_px and w // 3 >= min_tile_px:
        return "tiles_3x3", (h // 3, w // 3)
    warnings.warn(f"Image too small for tiling: shape=({h}, {w}), min_tile_px={min_tile_px}.",
                  RuntimeWarning, stacklevel=2)
    return "off", None


def tiles_meta(h, w, *, tile_mode, tile_shape_px=None):
    """metrics/common.py:173-217."""
    meta = {"tile_mode": tile_mode}
    if tile_mode == "off":
        return meta
    if tile_shape_px is None:
        raise ValueError("tile_shape_px must be provided when tile_mode is not 'off'.")
    meta.update({"tile_grid_shape": (3, 3), "tile_labels": TILE_LABELS, "tile_order": "row-major",
                 "tile_shape_px": (int(tile_shape_px[0]), int(tile_shape_px[1])),
                 "used_subtiles": bool(tile_mode == "subtiles_9x9")})
    return meta


def tiled_scalar_fields(image, *, tile_mode, compute_fn):
    """metrics/common.py:248-378 -- 3x3 direct tiles (std = NaN) or 9x9 sub-tiles
    aggregated to 3x3 mean / population std."""
    img = np.asarray(image)
    if img.ndim != 2:
        raise ValueError(f"tiled_scalar_fields expects a 2D array, got ndim={img.ndim}")
    n = {"tiles_3x3": 3, "subtiles_9x9": 9}.get(tile_mode)
    if n is None:
        raise ValueError("tile_mode must be 'tiles_3x3' or 'subtiles_9x9'.")
    ye, xe = split_edges(img.shape[0], n), split_edges(img.shape[1], n)
    grids = None
    for r in range(n):
        for c in range(n):
            vals = compute_fn(img[ye[r][0]:ye[r][1], xe[c][0]:xe[c][1]])
            if grids is None:
                if not vals:
                    raise ValueError("compute_fn returned an empty dict for the first tile.")
                grids = {k: np.empty((n, n), dtype=float) for k in vals}
            for k in grids:
                grids[k][r, c] = float(vals[k])
    out = {}
    for k, g in grids.items():
        if n == 3:
            out[k] = {"mean": np.asarray(g, dtype=float), "std": np.full((3, 3), np.nan)}
        else:
            mean = np.empty((3, 3))
            std = np.empty((3, 3))
            for r in range(3):
                for c in range(3):
                    b = g[3 * r:3 * r + 3, 3 * c:3 * c + 3]
                    mean[r, c] = float(np.mean(b))
                    std[r, c] = float(np.std(b, ddof=0))
            out[k] = {"mean": mean, "std": std}
    return out


def stack_time_series(values):
    """metrics/common.py:381-408."""
    if not values:
        raise ValueError("No values provided for stacking.")
    v0 = values[0]
    if isinstance(v0, dict):
        return {k: stack_time_series([v[k] for v in values]) for k in v0}
    if isinstance(v0, np.ndarray):
        return np.stack([np.asarray(v) for v in values], axis=0)
    if isinstance(v0, (float, int, np.floating, np.integer, bool, np.bool_)):
        return np.asarray(values)
    return list(values)


def normalize_groups(groups, *, all_groups, context, param_name="metrics"):
    """metrics/common.py:411-464."""
    if isinstance(groups, str):
        keys = {g.strip() for g in groups.split(",")} if "," in groups else {groups.strip()}
    elif hasattr(groups, "__iter__") and hasattr(groups, "__len__") and hasattr(groups, "__getitem__"):
        keys = set()
        for g in groups:
            if not isinstance(g, str):
                raise TypeError(f"{context}: {param_name} must be str or a sequence of str")
            keys.add(g.strip())
    else:
        raise TypeError(f"{context}: {param_name} must be str or a sequence of str")
    if "all" in keys:
        return set(all_groups)
    bad = sorted(k for k in keys if k not in all_groups)
    if bad:
        raise ValueError(f"{context}: unknown {param_name} group(s): {', '.join(bad)}. "
                         f"Allowed: {', '.join(sorted(all_groups))}")
    return keys


# --------------------------------------------------------------------------- aggregators
SPECKLE_GROUPS = {"amplitude", "grain", "bandwidth", "stats"}
SHARPNESS_GROUPS = {"stats", "gradient", "laplacian", "spectral", "autocorrelation", "eigenvalues"}


def _aggregate(image, kind, groups_all, metrics, tiles, display_origin, full_fns, tile_fns):
    if not isinstance(image, np.ndarray):
        raise TypeError(f"{kind}_stats expects a numpy.ndarray")
    if image.ndim != 2:
        raise ValueError(f"Expected 2D array, got ndim={image.ndim}")
    image = apply_display_origin(image, display_origin=display_origin)
    h, w = image.shape
    groups = normalize_groups(metrics, all_groups=groups_all, context=kind)
    out = {"meta": {"kind": kind, "display_origin": display_origin, "input_shape": (int(h), int(w)),
                    "requested_groups": sorted(groups)}, "full": {}}
    for g, fn in full_fns:
        if g in groups:
            out["full"][g] = fn(image)
    mode, tshape = choose_tiling_mode(h, w, tiles=tiles, min_tile_px=128)
    if mode == "off":
        return out
    out["meta"].update(tiles_meta(h, w, tile_mode=mode, tile_shape_px=tshape))
    t_out = {}
    for g, fn in tile_fns:
        if g in groups:
            t_out[g] = tiled_scalar_fields(image, tile_mode=mode, compute_fn=fn)
    if t_out:
        out["tiles"] = t_out
    return out


def speckle_stats(image, *, metrics="all", tiles=True, display_origin="lower",
                  saturation_value=65535.0, eps=1e-6, verbose=False):
    """metrics/speckles.py:83-255 (group order amplitude, grain, stats, bandwidth)."""
    dm = lambda a: distribution_moments(a, saturation_value=saturation_value, eps=eps)  # noqa: E731
    sub = lambda fn, ks: (lambda t: (lambda d: {k: float(d[k]) for k in ks})(fn(t)))  # noqa: E731
    full = [("amplitude", amplitude), ("grain", grain), ("stats", dm), ("bandwidth", bandwidth)]
    til = [("amplitude", sub(amplitude, ("visibility", "contrast"))),
           ("grain", sub(grain, ("lx", "ly", "leq", "r"))),
           ("stats", lambda t: {k: float(v) for k, v in dm(t).items()}),
           ("bandwidth", sub(bandwidth, ("spr", "feq", "f95", "sig_fx", "sig_fy", "rf")))]
    out = _aggregate(image, "speckles", SPECKLE_GROUPS, metrics, tiles, display_origin, full, til)
    return out


def sharpness_stats(image, *, metrics="all", tiles=True, display_origin="lower",
                    saturation_value=65535.0, eps=1e-6, verbose=False):
    """metrics/sharpness.py:89-288."""
    dm = lambda a: distribution_moments(a, saturation_value=saturation_value, eps=eps)  # noqa: E731
    sub = lambda fn, ks: (lambda t: (lambda d: {k: float(d[k]) for k in ks})(fn(t)))  # noqa: E731
    full = [("stats", dm), ("gradient", tenengrad),
            ("laplacian", lambda a: {"laplacian_variance": laplacian_variance(a)}),
            ("spectral", lambda a: {"spectral_entropy": spectral_entropy(a)}),
            ("autocorrelation", inverse_autocorr_width), ("eigenvalues", eigenvalues)]
    til = [("stats", lambda t: {k: float(v) for k, v in dm(t).items()}),
           ("gradient", sub(tenengrad, ("tenengrad", "ex", "ey", "re"))),
           ("laplacian", lambda t: {"laplacian_variance": float(laplacian_variance(t))}),
           ("spectral", lambda t: {"spectral_entropy": float(spectral_entropy(t))}),
           ("autocorrelation", sub(inverse_autocorr_width, ("sx", "sy", "seq", "r"))),
           ("eigenvalues", sub(eigenvalues, ("eigenvalues", "e1", "e2", "re")))]
    return _aggregate(image, "sharpness", SHARPNESS_GROUPS, metrics, tiles, display_origin, full, til)


def sharpness_stack_stats(stack, *, metrics="all", tiles=True, display_origin="lower",
                          saturation_value=65535.0, eps=1e-6, verbose=False, parallel=False,
                          n_jobs=None):
    """metrics/sharpness.py:290-399 (serial here; threads change nothing numerically)."""
    if not isinstance(stack, np.ndarray):
        raise TypeError("sharpness_stack_stats expects a numpy.ndarray")
    if stack.ndim != 3:
        raise ValueError(f"stack must be a 3D array with shape (T, H, W); got ndim={stack.ndim}")
    T, H, W = map(int, stack.shape)
    if T < 1:
        raise ValueError("stack must contain at least one frame.")
    per = [sharpness_stats(stack[t], metrics=metrics, tiles=tiles, display_origin=display_origin,
                           saturation_value=saturation_value, eps=eps) for t in range(T)]
    out = {"meta": {"kind": "sharpness_stack_stats", "input_shape": (H, W), "stack_shape": (T, H, W),
                    "n_frames": T, "display_origin": display_origin},
           "full": stack_time_series([d["full"] for d in per])}
    if tiles and all(isinstance(d.get("tiles"), dict) for d in per):
        out["tiles"] = stack_time_series([d["tiles"] for d in per])
    return out


def speckle_stack_stats(stack, *, metrics="all", tiles=True, display_origin="lower",
                        roi_grain_factor=3.0, roi_step_factor=0.5, tracking_method="template",
                        tracking_backend="skimage", subpixel=True, saturation_value=65535.0,
                        eps=1e-6, verbose=False, parallel=False, n_jobs=None):
    """metrics/speckles.py:258-490 -- phase A per-frame stats, then 3x3-ROI abs/inc tracking on
    the UN-flipped stack (the display-origin sign flip is commented out at 417-419)."""
    if not isinstance(stack, np.ndarray):
        raise TypeError("speckle_stack_stats expects a numpy.ndarray")
    if stack.ndim != 3:
        raise ValueError(f"stack must be a 3D array with shape (T, H, W); got ndim={stack.ndim}")
    T, H, W = map(int, stack.shape)
    if T < 1:
        raise ValueError("stack must contain at least one frame.")
    per = [speckle_stats(stack[t], metrics=metrics, tiles=tiles, display_origin=display_origin,
                         saturation_value=saturation_value, eps=eps) for t in range(T)]
    out_full = stack_time_series([d["full"] for d in per])
    out_tiles = None
    if tiles and all(isinstance(d.get("tiles"), dict) for d in per):
        out_tiles = stack_time_series([d["tiles"] for d in per])

    f0 = stack[0]
    g0 = grain(f0)
    ell = float(np.nanmax([g0["lx"], g0["ly"], g0["leq"]]))
    if not np.isfinite(ell) or ell <= 0:
        raise ValueError("Could not infer a valid grain size from frame 0 (lx/ly/leq).")
    side = odd_size(int(np.ceil(roi_grain_factor * ell)))
    step = int(max(1, round(roi_step_factor * side)))
    grid, labels = roi_grid_3x3((H, W), (side, side), (step, step), center_yx=None)

    d = {k: np.empty((T, 3, 3), dtype=np.float32) for k in ("dxa", "dya", "dxi", "dyi")}
    for t in range(T):
        cur = stack[t]
        prev = stack[t - 1] if t > 0 else stack[0]
        for iy in range(3):
            for ix in range(3):
                sy, sx = grid[iy, ix]
                kw = dict(slices_yx=(sy, sx), method=tracking_method, backend=tracking_backend,
                          subpixel=subpixel, eps=1e-9)
                dy_a, dx_a, _, _ = track_translation(f0[sy, sx], cur, **kw)
                dy_i, dx_i, _, _ = track_translation(prev[sy, sx], cur, **kw)
                d["dya"][t, iy, ix], d["dxa"][t, iy, ix] = dy_a, dx_a
                d["dyi"][t, iy, ix], d["dxi"][t, iy, ix] = dy_i, dx_i

    def block(dx, dy):
        r = np.sqrt(dx ** 2 + dy ** 2)
        f = lambda fn, a: fn(a, axis=(1, 2)).astype(np.float32)  # noqa: E731
        return {"dx": f(np.nanmean, dx), "dy": f(np.nanmean, dy), "r": f(np.nanmean, r),
                "std_dx": f(np.nanstd, dx), "std_dy": f(np.nanstd, dy), "std_r": f(np.nanstd, r)}

    out = {"meta": {"kind": "speckle_stack_stats", "input_shape": (H, W), "stack_shape": (T, H, W),
                    "n_frames": T, "display_origin": display_origin,
                    "grain0": {k: g0.get(k) for k in ("lx", "ly", "leq", "r")},
                    "tracking": {"method": str(tracking_method), "backend": str(tracking_backend),
                                 "subpixel": bool(subpixel), "roi_size_yx": (side, side),
                                 "roi_step_yx": (step, step), "roi_labels": labels}},
           "full": out_full,
           "temporal": {"abs": block(d["dxa"], d["dya"]), "inc": block(d["dxi"], d["dyi"]),
                        "qc": {"roi_grid_shape": (3, 3)}}}
    if out_tiles is not None:
        out["tiles"] = out_tiles
    return out
