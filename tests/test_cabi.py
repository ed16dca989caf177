"""CPU tier: the C-ABI library builds/loads and exports every symbol include/b4d.h declares.
No compute call is made here (no GPU in this tier)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from barc4dip_amd import _ffi

    if not os.path.exists(_ffi.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "barc4dip_amd", "csrc"), "libb4d.so"], check=True)
    return _ffi.load_library()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "b4d.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(b4d_[a-z0-9_]+)\s*\(", txt)))


def test_header_library_binding_agree(lib):
    from barc4dip_amd import _ffi

    syms = header_symbols()
    assert len(syms) >= 16
    assert sorted(_ffi.SIGNATURES) == syms, "ctypes table and include/b4d.h drifted apart"
    assert lib.b4d_missing_symbols == ()
    out = subprocess.run(["nm", "-D", "--defined-only", _ffi.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (b4d_[a-z0-9_]+)", out))
    assert set(syms) <= exported


def test_library_is_not_a_timing_only_build(lib):
    """B4D_EXP_* switches (wrong results on purpose, used to price one ingredient of a kernel in A/B runs) compile only
    under -DB4D_TIMING_ONLY (csrc/b4d_timing_only.hpp), which stamps b4d_version(): such a library is refused here."""
    v = lib.b4d_version()
    assert b"TIMING" not in v and b"EXP" not in v, v
    src = os.path.join(ROOT, "barc4dip_amd", "csrc")
    guard = open(os.path.join(src, "b4d_timing_only.hpp")).read()
    used = set()
    for f in os.listdir(src):
        if f.endswith((".hip", ".hpp")) and f != "b4d_timing_only.hpp":
            used |= set(re.findall(r"#\s*(?:ifdef|ifndef|if|elif)[^\n]*\b(B4D_EXP_[A-Z0-9_]+)", open(os.path.join(src, f)).read()))
    missing = [m for m in sorted(used) if m not in guard]
    assert missing == [], f"timing-only switches not covered by the B4D_TIMING_ONLY guard: {missing}"


def test_no_gpu_calls_needed_for_introspection(lib):
    assert lib.b4d_version().startswith(b"b4d ")
    assert lib.b4d_size_supported(2048, 2048) == 1
    assert lib.b4d_size_supported(64, 4096) == 1
    assert lib.b4d_size_supported(171, 170) == 1                       # DFT-matrix range (any side <= 512)
    assert lib.b4d_size_supported(2160, 2560) == 1 and lib.b4d_size_supported(100, 2048) == 1   # 2^k * A * B splits
    assert lib.b4d_size_supported(1042, 2048) == 1                     # 1042 = 2 * 521: Bluestein
    assert lib.b4d_size_supported(4099, 2048) == 0                     # prime beyond the Bluestein range
    assert lib.b4d_size_supported(9000, 64) == 0 and lib.b4d_size_supported(1, 64) == 0
    assert lib.b4d_plan_destroy(None) == 0


def test_product_path_fails_loudly_without_gpu_or_library(tmp_path, monkeypatch):
    """No CPU fallback: a missing extension or a missing GPU raises, it never computes on the host."""
    import numpy as np
    import torch

    from barc4dip_amd import _ffi, signal

    with pytest.raises(_ffi.B4DUnavailable):
        _ffi.load_library(str(tmp_path / "nope.so"))
    if not torch.cuda.is_available():
        img = np.zeros((512, 512), dtype=np.float32)
        for fn in (lambda: signal.fft2d(img), lambda: signal.psd2d(img), lambda: signal.autocorr2d(img),
                   lambda: signal.xcorr2d(img, img), lambda: signal.phase_correlation(img[:31, :31], img)):
            with pytest.raises(_ffi.B4DUnavailable):
                fn()


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "barc4dip_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M):
                    bad.append(os.path.join(dirpath, f))
    assert bad == []
