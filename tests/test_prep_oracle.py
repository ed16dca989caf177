"""CPU tier: the flat-field oracle against the golden vectors the reference produced (tests/golden/prep.npz)."""
import os

import numpy as np
import pytest

from oracle import preprocess_np as P

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "prep.npz"))


def cases():
    im, fl, dk = G["imgs"], G["flats"], G["darks"]
    return {
        "default": (im, dict(flats=fl, darks=dk)),
        "mean": (im, dict(flats=fl, darks=dk, scale="flat_mean")),
        "none": (im, dict(flats=fl, darks=dk, scale="none")),
        "repair": (im, dict(flats=fl, darks=dk, bad_pixel_removal=True)),
        "eps50": (im, dict(flats=fl, darks=dk, eps=1500.0, bad_pixel_removal=True)),
        "single": (im[1], dict(flats=fl[0], darks=dk[0], bad_pixel_removal=True)),
        "flat_only": (im, dict(flats=fl)),
        "dark_only": (im, dict(darks=dk)),
        "neither": (im, {}),
        "f32_in": (im.astype(np.float32) * 0.37, dict(flats=fl.astype(np.float64), darks=dk[0])),
    }


@pytest.mark.parametrize("name", list(cases()))
def test_oracle_matches_reference_bitwise(name):
    img, kw = cases()[name]
    out = P.flat_field_correction(img, **kw)
    assert out.dtype == np.float32 and out.shape == G[name].shape
    assert np.array_equal(out, G[name], equal_nan=True)


def test_oracle_errors():
    im = G["imgs"]
    with pytest.raises(ValueError):
        P.flat_field_correction(im, flats=G["flats"], scale="median")
    with pytest.raises(ValueError):
        P.flat_field_correction(im[0, 0], flats=G["flats"])
    with pytest.raises(ValueError):
        P.flat_field_correction(im, flats=G["flats"][None])
