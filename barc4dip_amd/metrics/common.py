"""Tiling policy and result stacking shared by the aggregators (host side).

Same behaviour as ``barc4dip.metrics.common`` (common.py:44-464): display-origin flip, 3x3 tiles or 9x9
sub-tiles (>= 128 px per tile), round(linspace) edges, row-major NW..SE labels, (T,)-stacking of results."""
from __future__ import annotations

import warnings
from typing import Callable, Sequence

import numpy as np

TILE_GRID_SHAPE_3X3 = (3, 3)
TILE_ORDER = "row-major"
TILE_LABELS_3X3 = np.array([["NW", "N", "NE"], ["W", "C", "E"], ["SW", "S", "SE"]], dtype=object)


def normalize_display_origin(display_origin: str) -> str:
    origin = str(display_origin).strip().lower()
    if origin not in ("upper", "lower"):
        raise ValueError("display_origin must be 'upper' or 'lower'.")
    return origin


def apply_display_origin(image: np.ndarray, *, display_origin: str) -> np.ndarray:
    """'lower' (detector convention) flips the rows before any metric is computed (common.py:44-72)."""
    img = np.asarray(image)
    if img.ndim != 2:
        raise ValueError(f"apply_display_origin expects a 2D array, got ndim={img.ndim}")
    return img[::-1, :] if normalize_display_origin(display_origin) == "lower" else img


def split_edges(length: int, n_parts: int) -> list[tuple[int, int]]:
    """(start, stop) of n_parts near-equal parts; edges are round(linspace) (common.py:75-106)."""
    if length < 1:
        raise ValueError("length must be >= 1.")
    if n_parts < 1:
        raise ValueError("n_parts must be >= 1.")
    marks = np.linspace(0, length, n_parts + 1)
    spans = []
    for i in range(n_parts):
        lo = int(round(float(marks[i])))
        spans.append((lo, max(int(round(float(marks[i + 1]))), lo + 1)))
    spans[-1] = (spans[-1][0], length)
    return spans


def choose_tiling_mode(h: int, w: int, *, tiles: bool = False, min_tile_px: int = 128):
    """'subtiles_9x9' if h//9, w//9 >= min_tile_px, else 'tiles_3x3' if h//3, w//3 >= min_tile_px, else 'off'
    with a RuntimeWarning (common.py:109-170)."""
    if h < 1 or w < 1:
        raise ValueError("Invalid image shape (h and w must be >= 1).")
    if min_tile_px < 1:
        raise ValueError("min_tile_px must be >= 1.")
    if not bool(tiles):
        return "off", None
    for mode, n in (("subtiles_9x9", 9), ("tiles_3x3", 3)):
        if h // n >= min_tile_px and w // n >= min_tile_px:
            return mode, (h // n, w // n)
    warnings.warn(f"Image too small for tiling: shape=({h}, {w}), min_tile_px={min_tile_px}.", RuntimeWarning,
                  stacklevel=2)
    return "off", None


def tiles_meta(h: int, w: int, *, tile_mode: str, tile_shape_px=None) -> dict:
    meta: dict = {"tile_mode": tile_mode}
    if tile_mode == "off":
        return meta
    if tile_shape_px is None:
        raise ValueError("tile_shape_px must be provided when tile_mode is not 'off'.")
    meta.update({"tile_grid_shape": TILE_GRID_SHAPE_3X3, "tile_labels": TILE_LABELS_3X3, "tile_order": TILE_ORDER,
                 "tile_shape_px": (int(tile_shape_px[0]), int(tile_shape_px[1])),
                 "used_subtiles": bool(tile_mode == "subtiles_9x9")})
    return meta


def nan_std_grid_3x3() -> np.ndarray:
    return np.full((3, 3), np.nan, dtype=float)


def pack_mean_std(mean, std) -> dict:
    return {"mean": np.asarray(mean, dtype=float), "std": np.asarray(std, dtype=float)}


def aggregate_subtiles_9x9_to_3x3(sub):
    """Mean and population std of each 3x3 block of a 9x9 grid (common.py:248-275)."""
    arr = np.asarray(sub, dtype=float)
    if arr.shape != (9, 9):
        raise ValueError("Expected subtiles grid of shape (9, 9).")
    # block (r, c) = rows 3r..3r+2, columns 3c..3c+2 -> one (3, 3, 9) view, two reductions instead of eighteen (the stack
    # aggregators call this once per field, tile group and frame: it was a fifth of their host time)
    blocks = arr.reshape(3, 3, 3, 3).transpose(0, 2, 1, 3).reshape(3, 3, 9)
    return blocks.mean(axis=-1), blocks.std(axis=-1)


def tile_spans(h: int, w: int, tile_mode: str):
    n = {"tiles_3x3": 3, "subtiles_9x9": 9}.get(tile_mode)
    if n is None:
        raise ValueError("tile_mode must be 'tiles_3x3' or 'subtiles_9x9'.")
    return n, split_edges(h, n), split_edges(w, n)


def grids_to_fields(grids: dict[str, np.ndarray], n: int) -> dict[str, dict[str, np.ndarray]]:
    out = {}
    for k, g in grids.items():
        if n == 3:
            out[k] = pack_mean_std(g, nan_std_grid_3x3())
        else:
            out[k] = pack_mean_std(*aggregate_subtiles_9x9_to_3x3(g))
    return out


def tiled_scalar_fields(image, *, tile_mode: str, compute_fn: Callable[[np.ndarray], dict[str, float]]):
    """Run compute_fn on every tile -> {key: {"mean": (3,3), "std": (3,3)}} (common.py:278-378)."""
    img = image if hasattr(image, "device") else np.asarray(image)   # device tensors are sliced in place
    if img.ndim != 2:
        raise ValueError(f"tiled_scalar_fields expects a 2D array, got ndim={img.ndim}")
    n, ys, xs = tile_spans(int(img.shape[0]), int(img.shape[1]), tile_mode)
    grids = None
    for r, (y0, y1) in enumerate(ys):
        for c, (x0, x1) in enumerate(xs):
            vals = compute_fn(img[y0:y1, x0:x1])
            if grids is None:
                if not vals:
                    raise ValueError("compute_fn returned an empty dict for the first tile.")
                grids = {k: np.empty((n, n), dtype=float) for k in vals}
            for k in grids:
                grids[k][r, c] = float(vals[k])
    return grids_to_fields(grids, n)


def stack_time_series(values: list):
    """dicts of scalars -> dicts of (T,) arrays, arrays -> (T, ...) (common.py:381-408)."""
    if not values:
        raise ValueError("No values provided for stacking.")
    first = values[0]
    if isinstance(first, dict):
        return {k: stack_time_series([v[k] for v in values]) for k in first}
    if isinstance(first, np.ndarray):
        return np.stack([np.asarray(v) for v in values], axis=0)
    if isinstance(first, (float, int, np.floating, np.integer, bool, np.bool_)):
        return np.asarray(values)
    return list(values)


def normalize_groups(groups, *, all_groups: set[str], context: str, param_name: str = "metrics") -> set[str]:
    """'all', a comma-separated string or a sequence of group names -> validated set (common.py:411-464)."""
    if isinstance(groups, str):
        keys = {g.strip() for g in groups.split(",")} if "," in groups else {groups.strip()}
    elif isinstance(groups, Sequence):
        keys = set()
        for g in groups:
            if not isinstance(g, str):
                raise TypeError(f"{context}: {param_name} must be str or a sequence of str")
            keys.add(g.strip())
    else:
        raise TypeError(f"{context}: {param_name} must be str or a sequence of str")
    if "all" in keys:
        return set(all_groups)
    unknown = sorted(k for k in keys if k not in all_groups)
    if unknown:
        raise ValueError(f"{context}: unknown {param_name} group(s): {', '.join(unknown)}. "
                         f"Allowed: {', '.join(sorted(all_groups))}")
    return keys
