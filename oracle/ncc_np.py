"""Oracle (test infrastructure, NOT product code): ``barc4dip.signal.tracking.template_matching``
(SURVEY.md §8 row a10, §8f #3).

PARITY UNPINNED.  The reference's own code (signal/tracking.py:81-188, read as text) only prepares the inputs
(z-scored float32 template; z-scored float32 image for "opencv", raw float32 image for "skimage"), takes the arg-max
of the match map and converts it to a centre-to-centre shift.  The match map itself comes from
``cv2.matchTemplate(..., TM_CCOEFF_NORMED)`` or ``skimage.feature.match_template(pad_input=False)``; both libraries
are listed un-pinned in the reference's pyproject.toml:19-28 and are absent from this image, and the reference holds
no vectors for this path.  Both compute the zero-mean normalised cross-correlation over the "valid" positions,

    ncc[i, j] = sum_w (I[i+p, j+q] - mean_w I) (T[p, q] - mean T) / sqrt( sum_w (I - mean_w I)^2 * sum (T - mean T)^2 ),

restated here in float64 (window sums from summed-area tables, the correlation term from an exact FFT-free
``scipy.signal.correlate`` for small cases or FFT for large ones).  Following scikit-image's published implementation
the response is 0 where the denominator is not above float32 eps.  Self-checks (tests/test_ncc_oracle.py): brute-force
triple loop on small arrays, exact recovery of integer shifts, NCC = 1 at the true position.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import correlate

from . import signal_np as S


def match_template_ncc(image, template, *, method="auto"):
    """Zero-mean NCC map of shape (H-h+1, W-w+1), float32 (float64 arithmetic inside)."""
    img = np.asarray(image, dtype=np.float64)
    tpl = np.asarray(template, dtype=np.float64)
    H, W = img.shape
    h, w = tpl.shape
    vol = h * w
    tmean = tpl.mean()
    tssd = float(((tpl - tmean) ** 2).sum())
    sat1 = np.zeros((H + 1, W + 1))
    sat2 = np.zeros((H + 1, W + 1))
    sat1[1:, 1:] = img.cumsum(0).cumsum(1)
    sat2[1:, 1:] = (img * img).cumsum(0).cumsum(1)

    def wsum(s):
        return s[h:, w:] - s[:-h, w:] - s[h:, :-w] + s[:-h, :-w]

    s1, s2 = wsum(sat1), wsum(sat2)
    xc = correlate(img, tpl, mode="valid", method=method)
    num = xc - s1 * tmean
    den = (s2 - s1 * s1 / vol) * tssd
    den = np.sqrt(np.maximum(den, 0.0))
    out = np.zeros_like(xc)
    mask = den > np.finfo(np.float32).eps
    out[mask] = num[mask] / den[mask]
    return out.astype(np.float32)


def match_template_bruteforce(image, template):
    """The definition, loop by loop (small arrays only)."""
    img = np.asarray(image, dtype=np.float64)
    tpl = np.asarray(template, dtype=np.float64)
    H, W = img.shape
    h, w = tpl.shape
    t0 = tpl - tpl.mean()
    tn = np.sqrt((t0 * t0).sum())
    out = np.zeros((H - h + 1, W - w + 1))
    for i in range(H - h + 1):
        for j in range(W - w + 1):
            win = img[i:i + h, j:j + w]
            w0 = win - win.mean()
            d = np.sqrt((w0 * w0).sum()) * tn
            out[i, j] = (w0 * t0).sum() / d if d > np.finfo(np.float32).eps else 0.0
    return out.astype(np.float32)


def template_matching(template, image, *, slices_yx=None, backend="opencv", subpixel=True, eps=1e-9):
    """signal/tracking.py:81-188 with the third-party match map replaced by match_template_ncc."""
    tpl = S.as_float2d(template, "template")
    img = S.as_float2d(image, "image")
    H, W = img.shape
    h, w = tpl.shape
    if h > H or w > W:                                                    # :134-135
        raise ValueError(f"template shape {(h, w)} must fit inside image shape {(H, W)}")
    if slices_yx is None:                                                 # :137-138
        slices_yx = S.roi_slices((H, W), (h, w), center_yx=None, clip=False)
    sy, sx = slices_yx
    y0 = (sy.start + sy.stop - 1) / 2.0                                   # :142-143
    x0 = (sx.start + sx.stop - 1) / 2.0
    tpl_z = S.zscore2d(tpl, eps).astype(np.float32, copy=False)           # :145
    if backend == "opencv":                                               # :147-158
        corr = match_template_ncc(S.zscore2d(img, eps).astype(np.float32, copy=False), tpl_z)
    elif backend == "skimage":                                            # :160-167
        corr = match_template_ncc(img.astype(np.float32, copy=False), tpl_z)
    else:
        raise ValueError("backend must be 'opencv' or 'skimage'.")
    mi, mj = np.unravel_index(int(np.argmax(corr)), corr.shape)           # :172
    peak, snr = S.corr_peak_quality(corr, (mi, mj), eps)
    py, px = float(mi), float(mj)
    if subpixel:                                                          # :177-180
        di, dj = S.peak_subpixel_taylor(corr, (mi, mj))
        py += float(di)
        px += float(dj)
    return float(py + (h - 1) / 2.0 - y0), float(px + (w - 1) / 2.0 - x0), float(peak), float(snr)
