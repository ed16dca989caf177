"""Generate tests/golden/prep.npz by RUNNING THE REAL REFERENCE's flat_field_correction (build container only).

    python oracle/make_golden_prep.py

Data only: seeded synthetic inputs and the outputs the reference produced for them.
"""
from __future__ import annotations

import importlib
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import load_reference  # noqa: E402


def inputs(seed=7, t=4, h=48, w=64):
    """Detector-like data: uint16 counts, a smooth gain pattern, dark offsets, a few dead / hot pixels."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    gain = 1.0 + 0.3 * np.sin(xx / 9.0) * np.cos(yy / 7.0)
    dark_level = 100 + 5 * rng.random((h, w))
    flats = rng.poisson(2000 * gain[None] + dark_level[None], size=(5, h, w)).astype(np.uint16)
    darks = rng.poisson(dark_level[None], size=(3, h, w)).astype(np.uint16)
    imgs = rng.poisson(800 * gain[None] * (1 + 0.5 * rng.random((t, h, w))) + dark_level[None]).astype(np.uint16)
    dead = [(3, 5), (3, 6), (20, 0), (47, 63), (0, 0), (30, 31)]
    for (i, j) in dead:
        flats[:, i, j] = darks[:, i, j].mean(axis=0).astype(np.uint16)   # F - D <= 0 (or tiny)
    flats[:, 10, 10] = 0                                                # F < D
    return imgs, flats, darks


def main():
    load_reference.load()
    norm = importlib.import_module("barc4dip.preprocessing.normalize")
    imgs, flats, darks = inputs()
    g = {"versions": np.array([np.__version__, scipy.__version__]), "imgs": imgs, "flats": flats, "darks": darks}
    f = norm.flat_field_correction
    g["default"] = f(imgs, flats=flats, darks=darks)
    g["mean"] = f(imgs, flats=flats, darks=darks, scale="flat_mean")
    g["none"] = f(imgs, flats=flats, darks=darks, scale="none")
    g["repair"] = f(imgs, flats=flats, darks=darks, bad_pixel_removal=True)
    g["eps50"] = f(imgs, flats=flats, darks=darks, eps=1500.0, bad_pixel_removal=True)
    g["single"] = f(imgs[1], flats=flats[0], darks=darks[0], bad_pixel_removal=True)
    g["flat_only"] = f(imgs, flats=flats)
    g["dark_only"] = f(imgs, darks=darks)
    g["neither"] = f(imgs)
    g["f32_in"] = f(imgs.astype(np.float32) * 0.37, flats=flats.astype(np.float64), darks=darks[0])
    out = os.path.join(ROOT, "tests", "golden", "prep.npz")
    np.savez_compressed(out, **g)
    print(out, os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
