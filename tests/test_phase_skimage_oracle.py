"""CPU tier: self-checks of oracle/phase_skimage_np.py -- the published algorithm of skimage.registration.phase_cross_correlation
behind ``phase_correlation(backend="skimage")`` (signal/tracking.py:262-272).  Parity with scikit-image itself is UNPINNED (the
package is absent from the image, the reference holds no vectors): these tests hold the restatement to what the algorithm must
do, and to the internal back-end where the two coincide."""
import numpy as np
import pytest

from barc4dip_amd import synth
from oracle import phase_skimage_np as P
from oracle import signal_np as S


def _fourier_shift(img, dy, dx):
    """Circular sub-pixel shift by the Fourier shift theorem (float64)."""
    ny, nx = img.shape
    ky = np.fft.fftfreq(ny)[:, None]
    kx = np.fft.fftfreq(nx)[None, :]
    return np.real(np.fft.ifft2(np.fft.fft2(img) * np.exp(-2j * np.pi * (ky * dy + kx * dx))))


def test_integer_shifts_are_exact_and_match_the_internal_backend():
    base = synth.speckle_frame(256, 7).astype(np.float64)
    sl = (slice(60, 181), slice(70, 191))
    for dy, dx in ((3, -5), (-17, 22), (0, 0), (40, 40)):
        fr = np.roll(base, (dy, dx), axis=(0, 1))
        got = P.phase_correlation_skimage(base[sl], fr, slices_yx=sl, subpixel=False)
        assert (got[0], got[1]) == (dy, dx) and np.isnan(got[2]) and np.isnan(got[3])
        ref = S.phase_correlation(base[sl], fr, slices_yx=sl, subpixel=False)
        assert (got[0], got[1]) == (ref[0], ref[1])
        up = P.phase_correlation_skimage(base[sl], fr, slices_yx=sl, subpixel=True)
        assert abs(up[0] - dy) <= 0.05 + 1e-12 and abs(up[1] - dx) <= 0.05 + 1e-12      # on the 0.1-px grid around the true shift


@pytest.mark.parametrize("shift", [(2.3, -4.6), (-0.4, 0.7), (10.5, 3.25)])
def test_fractional_shifts_on_the_tenth_pixel_grid(shift):
    base = synth.speckle_frame(128, 11).astype(np.float64)
    moved = _fourier_shift(base, *shift)
    got = P.phase_cross_correlation(moved, base, upsample_factor=10)
    assert np.allclose(got * 10, np.round(got * 10))                                  # multiples of 0.1
    assert abs(got[0] - shift[0]) <= 0.1 and abs(got[1] - shift[1]) <= 0.1
    got100 = P.phase_cross_correlation(moved, base, upsample_factor=100)
    assert abs(got100[0] - shift[0]) <= 0.02 and abs(got100[1] - shift[1]) <= 0.02


def test_upsampled_dft_equals_the_dense_transform():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(12, 10)) + 1j * rng.normal(size=(12, 10))
    # up-sampling factor 1, region = the whole array, no offset: the plain 2-D DFT
    assert np.allclose(P.upsampled_dft(x, 12, 1, (0, 0))[:, :10], np.fft.fft(np.fft.fft(x, axis=1), axis=0)[:, :10])
