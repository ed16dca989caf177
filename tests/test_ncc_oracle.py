"""CPU tier: self-checks of the NCC template-matching oracle (parity unpinned: cv2 / scikit-image are absent)."""
import numpy as np
import pytest

from barc4dip_amd import synth
from oracle import ncc_np as N


def test_fast_ncc_equals_bruteforce():
    rng = np.random.default_rng(5)
    img = rng.normal(size=(23, 31)) * 3 + 10
    tpl = img[4:11, 9:18] + rng.normal(size=(7, 9)) * 0.1
    a = N.match_template_ncc(img, tpl)
    b = N.match_template_bruteforce(img, tpl)
    assert a.shape == (17, 23)
    np.testing.assert_allclose(a, b, atol=2e-6)
    assert np.unravel_index(np.argmax(a), a.shape) == (4, 9)


def test_constant_window_gives_zero_response():
    img = np.zeros((16, 16))
    img[8:, :] = np.arange(8 * 16).reshape(8, 16)
    tpl = np.arange(12.0).reshape(3, 4)
    out = N.match_template_ncc(img, tpl)
    assert np.all(out[:6, :] == 0.0)
    np.testing.assert_allclose(out, N.match_template_bruteforce(img, tpl), atol=2e-6)


@pytest.mark.parametrize("backend", ["opencv", "skimage"])
def test_integer_shift_recovered(backend):
    n, h = 128, 33
    f0 = synth.speckle_frame(n, 3)
    sy, sx = slice(40, 40 + h), slice(50, 50 + h)
    for (dy, dx) in ((0, 0), (5, -7), (-12, 9)):
        f1 = np.roll(f0, (dy, dx), axis=(0, 1))
        r = N.template_matching(f0[sy, sx], f1, slices_yx=(sy, sx), backend=backend, subpixel=False)
        assert (r[0], r[1]) == (dy, dx)
        assert r[2] == pytest.approx(1.0, abs=1e-5)
        r2 = N.template_matching(f0[sy, sx], f1, slices_yx=(sy, sx), backend=backend, subpixel=True)
        assert abs(r2[0] - dy) < 0.05 and abs(r2[1] - dx) < 0.05


def test_errors():
    a = np.zeros((8, 8))
    with pytest.raises(ValueError):
        N.template_matching(np.zeros((9, 4)), a)
    with pytest.raises(ValueError):
        N.template_matching(np.zeros((3, 3)), a, backend="internal")
    with pytest.raises(ValueError):
        N.template_matching(np.zeros((3,)), a)
