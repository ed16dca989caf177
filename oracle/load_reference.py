"""Build-container-only helper: import the real barc4dip sub-packages from
/root/reference WITHOUT executing barc4dip/__init__.py (which needs h5py).

Used only by oracle/make_golden.py (fixture generation) and tests marked
``needs_reference`` (skipped wherever /root/reference is absent, e.g. the GPU box).
Never imported by product code, bench.py or the -m gpu tests.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF_ROOT = "/root/reference/src/barc4dip"


def available() -> bool:
    return os.path.isdir(REF_ROOT)


def load():
    """Return a namespace with the reference's signal / metrics / geometry / maths modules."""
    if not available():
        raise RuntimeError("reference tree not present")
    sys.dont_write_bytecode = True            # the reference mount is read-only
    if "barc4dip" not in sys.modules:
        pkg = types.ModuleType("barc4dip")
        pkg.__path__ = [REF_ROOT]
        sys.modules["barc4dip"] = pkg
    ns = types.SimpleNamespace()
    ns.signal = importlib.import_module("barc4dip.signal")
    ns.tracking = importlib.import_module("barc4dip.signal.tracking")
    ns.corr = importlib.import_module("barc4dip.signal.corr")
    ns.fft = importlib.import_module("barc4dip.signal.fft")
    ns.metrics = importlib.import_module("barc4dip.metrics")
    ns.speckles = importlib.import_module("barc4dip.metrics.speckles")
    ns.sharpness = importlib.import_module("barc4dip.metrics.sharpness")
    ns.common = importlib.import_module("barc4dip.metrics.common")
    ns.roi = importlib.import_module("barc4dip.geometry.roi")
    ns.masks = importlib.import_module("barc4dip.geometry.masks")
    ns.mstats = importlib.import_module("barc4dip.maths.stats")
    ns.radial = importlib.import_module("barc4dip.maths.radial")
    ns.filters = importlib.import_module("barc4dip.preprocessing.filters")
    return ns
