"""ctypes binding of include/b4d.h (the C ABI of the gfx950 kernels).

The product path has no CPU fallback: if the shared library is missing or a GPU is not
present, every compute entry point raises (`B4DUnavailable`).  Build with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C barc4dip_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libb4d.so")

REMOVE_MEAN = 1
NORM_PEAK = 2
STANDARDIZE = 4


class B4DUnavailable(RuntimeError):
    """The HIP extension (libb4d.so) or the GPU is missing; there is no CPU fallback."""


class B4DError(RuntimeError):
    pass


class B4DSizeError(B4DError, NotImplementedError):
    """(ny, nx) has no native plan."""


_lib = None
_lock = threading.Lock()

_vp, _i, _u, _f, _d, _sz = C.c_void_p, C.c_int, C.c_uint, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); every symbol declared in include/b4d.h
SIGNATURES = {
    "b4d_version": (C.c_char_p, []),
    "b4d_last_error": (C.c_char_p, []),
    "b4d_set_option": (_i, [C.c_char_p, _i]),
    "b4d_size_supported": (_i, [_i, _i]),
    "b4d_plan_create_general": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "b4d_fft2d_c2c": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "b4d_plan_create": (_i, [_i, _i, _i, C.POINTER(_vp)]),
    "b4d_plan_destroy": (_i, [_vp]),
    "b4d_plan_workspace_bytes": (_sz, [_vp]),
    "b4d_fft2d": (_i, [_vp, _vp, _i, _vp, _vp]),
    "b4d_psd2d": (_i, [_vp, _vp, _i, _vp, _f, _vp]),
    "b4d_autocorr2d": (_i, [_vp, _vp, _i, _vp, _u, _vp]),
    "b4d_psd_autocorr2d": (_i, [_vp, _vp, _i, _vp, _f, _vp, _u, _vp]),
    "b4d_psd_autocorr2d_timed": (_i, [_vp, _vp, _i, _vp, _f, _vp, _u, _vp, _vp]),
    "b4d_plan_tune": (_i, [_vp, _vp, _i, _vp, _f, _vp, _u, _i, _vp, _vp, _vp]),
    "b4d_xcorr2d": (_i, [_vp, _vp, _vp, _i, _vp, _u, _vp]),
    "b4d_phase_correlation": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _d, _vp, _vp, _vp]),
    "b4d_template_match": (_i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp]),
    "b4d_temporal_accumulate": (_i, [_vp, _i, _sz, _vp, _vp, _vp]),
    "b4d_temporal_finalize": (_i, [_vp, _vp, _d, _sz, _vp, _vp, _vp, _vp]),
    "b4d_temporal_accumulate_range": (_i, [_vp, _i, _sz, _sz, _sz, _vp, _vp, _vp]),
    "b4d_temporal_finalize_dev": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "b4d_moments": (_i, [_vp, _i, _sz, _d, _d, _vp, _vp]),
    "b4d_sobel_laplace_stats": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "b4d_stack_mean_f32": (_i, [_vp, _i, _sz, _vp, _vp]),
    "b4d_flat_den": (_i, [_vp, _vp, _sz, _f, _i, _vp, _vp]),
    "b4d_flat_field": (_i, [_vp, _i, _sz, _vp, _vp, _f, _f, _i, _vp, _vp]),
    "b4d_to_f32": (_i, [_vp, _i, _sz, _vp, _vp]),
    "b4d_repair_pixels": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "b4d_sta2_eigenvalues": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
    "b4d_percentiles": (_i, [_vp, _i, _sz, _vp, _i, _vp, _vp]),
    "b4d_radial_profile": (_i, [_vp, _i, _i, _i, _i, _i, _d, _vp, _vp]),
    "b4d_psd_stats": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "b4d_richardson_lucy": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _f, _i, _vp, _vp]),
    "b4d_wiener_create": (_i, [_i, _i, _vp, _i, _i, _f, C.POINTER(_vp)]),
    "b4d_wiener_apply": (_i, [_vp, _vp, _i, _vp, _i, _vp]),
    "b4d_wiener_destroy": (_i, [_vp]),
    "b4d_uw_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_ulonglong, _i, _i, C.c_float, C.c_float, _i, _i, _vp, _vp]),
}


def load_library(path: str | None = None):
    """dlopen libb4d.so and attach prototypes.  Does not touch the GPU."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise B4DUnavailable(
                f"{p} not found: the HIP extension is not built (run __graft_entry__.build()). "
                "barc4dip_amd has no CPU fallback.")
        try:    # load PyTorch's HIP runtime first: libb4d.so must bind to the SAME libamdhip64 instance (a second copy of the
            import torch  # noqa: F401    # runtime, loaded by dlopen-ing us before torch, does not see the devices)
        except Exception:  # pragma: no cover - torch-less host tools still get the symbol table
            pass
        lib = C.CDLL(p)
        missing = []
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:        # header/library drift; tests assert this list is empty
                missing.append(name)
                continue
            fn.restype = res
            fn.argtypes = args
        lib.b4d_missing_symbols = tuple(missing)
        if path is None:
            _lib = lib
        return lib


def lib():
    return load_library()


def check(rc: int):
    if rc == 0:
        return
    msg = lib().b4d_last_error().decode("utf-8", "replace")
    if rc == -2:
        raise B4DSizeError(msg)
    raise B4DError(f"b4d error {rc}: {msg}")


def require_gpu():
    import torch

    if not torch.cuda.is_available():
        raise B4DUnavailable("no ROCm device visible: barc4dip_amd computes on the GPU only (no CPU fallback).")
    return torch


class Plan:
    """RAII wrapper of b4d_plan (twiddles + chunk workspace) for one (ny, nx)."""

    def __init__(self, ny: int, nx: int, chunk: int = 8, general: bool = False):
        require_gpu()
        self.ny, self.nx, self.chunk = int(ny), int(nx), int(chunk)
        h = _vp()
        create = lib().b4d_plan_create_general if general else lib().b4d_plan_create
        check(create(self.ny, self.nx, self.chunk, C.byref(h)))
        self._h = h

    @property
    def handle(self):
        return self._h

    def workspace_bytes(self) -> int:
        return int(lib().b4d_plan_workspace_bytes(self._h))

    def tune(self, frames, psd=None, autocorr=None, psd_scale: float = 1.0, flags: int = 0, candidates: int = 3):
        """b4d_plan_tune: keep the fastest of up to `candidates` workspace allocations for THIS call (device tensors
        `frames` (T, ny, nx) float32 and the output tensors of a normal psd / autocorrelation call).  Returns
        (ms per pass on the kept workspace, ms per pass on the slowest candidate).  Where a multi-GB buffer lands in device
        memory is worth 5-10 % of every kernel streaming through it."""
        best, worst = C.c_float(0.0), C.c_float(0.0)
        check(lib().b4d_plan_tune(self._h, C.c_void_p(frames.data_ptr()), int(frames.shape[0]),
                                  C.c_void_p(psd.data_ptr()) if psd is not None else None, float(psd_scale),
                                  C.c_void_p(autocorr.data_ptr()) if autocorr is not None else None, int(flags), int(candidates),
                                  C.byref(best), C.byref(worst), stream_ptr()))
        return float(best.value), float(worst.value)

    def close(self):
        if getattr(self, "_h", None):
            lib().b4d_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_plans: dict = {}
_MAX_CACHED_PLANS = 12
_plans_lock = threading.Lock()


def supported(ny: int, nx: int) -> bool:
    return bool(lib().b4d_size_supported(int(ny), int(nx)))


def default_chunk(ny: int, nx: int) -> int:
    """Frames per launch group.  The column kernel runs ONE workgroup per CU, so a launch needs many
    multiples of 256 column tiles to amortise its ragged tail: aim at >= 4096 tiles per launch
    (measured on MI355X at 2048^2: chunk 6 -> 21k frames/s, 32 -> 34k, 64 -> 36k), workspace <= 1 GiB."""
    pow2 = lambda n: 64 <= n <= 4096 and n & (n - 1) == 0  # noqa: E731
    if not (pow2(ny) and pow2(nx)):   # general-length plan: three complex chunk buffers, <= 1 GiB in total (2 GiB for
        # detector-sized frames: the persistent row kernels of the mixed-radix route want >= 16 frames per launch)
        budget = (2048 << 20) if max(ny, nx) > 1024 else (1024 << 20)
        return max(1, min(128, budget // (24 * ny * nx)))
    ct = 8 if ny == 4096 else 16
    tiles = max(1, (nx // 2) // ct)
    chunk = -(-4096 // tiles)
    cap = max(1, (1 << 30) // (ny * nx * 4))
    return max(1, min(chunk, cap, 128))


def stack_chunk(ny: int, nx: int, frames: int) -> int:
    """Launch group for the stack entry points (psd2d_stack / autocorr2d_stack / psd_autocorr2d_stack) on power-of-two frames:
    the default group, or up to four times as many frames when the stack is that long and the workspace stays under 4.5 GiB
    (2048^2: 64 -> 256 frames, 4.3 GB of a 288 GB device) -- the column pass runs 5-7 % faster on 16 384 tiles per launch than
    on 4 096.  Per-frame calls, tracking and general-size plans keep the default (their work buffers scale with it)."""
    base = default_chunk(ny, nx)
    pow2 = lambda n: 64 <= n <= 4096 and n & (n - 1) == 0  # noqa: E731
    if not (pow2(int(ny)) and pow2(int(nx))) or frames <= base or int(ny) * int(nx) < (1 << 20):
        return base     # (measured at 512^2: 666 k frames/s with the default 128, 645 k with 512; 1024^2: 171 -> 185 k; 4096^2: 9.0 -> 9.3 k)
    cap = max(1, int(4.5 * (1 << 30)) // (int(ny) * int(nx) * 4))
    for mult in (2, 4):     # two sizes beyond the default, so that stacks of every length share three cached plans
        if frames <= mult * base or mult == 4:
            return max(base, min(mult * base, cap))
    return base


def get_plan(ny: int, nx: int, chunk: int | None = None, general: bool = False) -> Plan:
    import torch

    if general and chunk is None:
        chunk = max(1, min(128, (256 << 20) // (24 * int(ny) * int(nx))))
    # plans own device workspace: one per (device, stream), so that work queued on different streams never shares it
    key = (int(ny), int(nx), int(chunk or default_chunk(ny, nx)), torch.cuda.current_device(),
           int(torch.cuda.current_stream().cuda_stream), bool(general))
    with _plans_lock:
        pl = _plans.pop(key, None)
        if pl is None:
            pl = Plan(key[0], key[1], key[2], general=general)
        _plans[key] = pl                       # most recently used last
        while len(_plans) > _MAX_CACHED_PLANS:  # plans own up to 1 GiB of workspace each: keep a bounded LRU set
            _plans.pop(next(iter(_plans)))     # dropped from the cache only; the plan dies with its last user
        return pl


def stream_ptr():
    import torch

    return _vp(torch.cuda.current_stream().cuda_stream)
