"""Radial reductions -- drop-in for ``barc4dip.maths.radial``.

``radial_mean_interpolated`` (radial.py:101-169: polar sampling + bilinear interpolation, the hot part of
`grain` / `inverse_autocorr_width`) runs on the GPU (b4d_radial_profile); ``radial_mean_binned``
(radial.py:38-98) is a NumPy bincount on the host (not on the measured path)."""
from __future__ import annotations

import numpy as np

from .. import _device as D
from .. import _ffi


def _pixel_axes(shape):
    ny, nx = shape
    return np.arange(nx, dtype=float) - (nx // 2), np.arange(ny, dtype=float) - (ny // 2)


def _rmax(shape, r_max):
    x, y = _pixel_axes(shape)
    if r_max is None:
        r_max = min(float(np.max(np.abs(x))), float(np.max(np.abs(y))))
    if r_max <= 0:
        raise ValueError("r_max must be > 0 (or leave it as None with valid shape).")
    return float(r_max)


def radial_mean_binned(signal_2d, *, r_max=None, bin_size: float = 1.0):
    z = D.to_host(signal_2d) if D.is_tensor(signal_2d) else np.asarray(signal_2d)
    z = np.asarray(z, dtype=float)
    if z.ndim != 2:
        raise ValueError("signal_2d must be a 2D array.")
    if not np.isfinite(z).all():
        raise ValueError("signal_2d contains non-finite values.")
    if bin_size <= 0:
        raise ValueError("bin_size must be > 0.")
    r_max = _rmax(z.shape, r_max)
    x, y = _pixel_axes(z.shape)
    rr = np.sqrt(x[None, :] ** 2 + y[:, None] ** 2)
    nbins = int(np.floor(r_max / bin_size)) + 1
    which = np.floor(rr / bin_size).astype(np.int64)
    inside = which < nbins
    tot = np.bincount(which[inside], weights=z[inside], minlength=nbins).astype(float)
    cnt = np.bincount(which[inside], minlength=nbins).astype(float)
    prof = np.full(nbins, np.nan)
    prof[cnt > 0] = tot[cnt > 0] / cnt[cnt > 0]
    return prof, (np.arange(nbins, dtype=float) + 0.5) * float(bin_size)


def radial_profile_batch(maps, *, r_max=None, nr=None, ntheta=None):
    """(B, ny, nx) maps (device or host) -> ((B, nr) float64 ndarray, r)."""
    torch = _ffi.require_gpu()
    t, _, _ = D.to_device_f32(maps, ndim=(3,))
    b, ny, nx = (int(v) for v in t.shape)
    r_max = _rmax((ny, nx), r_max)
    nr = int(np.floor(r_max)) + 1 if nr is None else int(nr)
    ntheta = int(2.0 * np.pi * 180.0) if ntheta is None else int(ntheta)
    if nr <= 1:
        raise ValueError("nr must be > 1.")
    if ntheta <= 3:
        raise ValueError("ntheta must be > 3.")
    out = torch.empty((b, nr), dtype=torch.float64, device=t.device)
    _ffi.check(_ffi.lib().b4d_radial_profile(D.ptr(t), b, ny, nx, nr, ntheta, float(r_max), D.ptr(out), _ffi.stream_ptr()))
    return out.cpu().numpy(), np.linspace(0.0, r_max, nr)


def radial_mean_interpolated(signal_2d, *, r_max=None, nr=None, ntheta=None, fill_value: float = 0.0):
    if not D.is_tensor(signal_2d):
        signal_2d = np.asarray(signal_2d)
        if signal_2d.ndim == 2 and not np.isfinite(signal_2d).all():
            raise ValueError("signal_2d contains non-finite values.")
    if signal_2d.ndim != 2:
        raise ValueError("signal_2d must be a 2D array.")
    prof, r = radial_profile_batch(signal_2d[None], r_max=r_max, nr=nr, ntheta=ntheta)
    if fill_value != 0.0:
        # The kernel takes samples outside the grid as 0 (RegularGridInterpolator(bounds_error=False, fill_value=0)); any other
        # fill value adds fill_value x (share of such samples) at every radius -- a function of the sampling geometry alone.
        ny, nx = (int(v) for v in signal_2d.shape)
        nt = int(2.0 * np.pi * 180.0) if ntheta is None else int(ntheta)
        th = np.arange(nt, dtype=float) * (2.0 * np.pi / nt)
        px, py = r[:, None] * np.cos(th)[None, :], r[:, None] * np.sin(th)[None, :]
        outside = (px < -(nx // 2)) | (px > nx - 1 - nx // 2) | (py < -(ny // 2)) | (py > ny - 1 - ny // 2)
        prof = prof + float(fill_value) * np.mean(outside, axis=1)[None, :]
    return prof[0], r
