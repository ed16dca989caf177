"""CPU tier: host-side logic of the product package (argument validation, geometry, sharding)."""
import numpy as np
import pytest

from barc4dip_amd import geometry as G
from barc4dip_amd.signal import common as C
from barc4dip_amd.signal import fft as F
from barc4dip_amd.signal import corr as X
from oracle import signal_np as S


def test_axes_match_oracle():
    for n, dx in ((8, 1.0), (9, 0.25), (2048, 3.0)):
        np.testing.assert_array_equal(F.freq_axis1d(n=n, dx=dx), S.freq_axis1d(n=n, dx=dx))
    x = np.linspace(0, 6.3, 64)
    y = np.linspace(5, -5, 33)
    a, b = F.freq_axes2d(shape=(33, 64), x=x, y=y)
    ra, rb = S.freq_axes2d(shape=(33, 64), x=x, y=y)
    np.testing.assert_array_equal(a, ra)
    np.testing.assert_array_equal(b, rb)
    np.testing.assert_array_equal(C._lag_axis_from_step(7, 0.5), S.lag_axis(7, 0.5))


@pytest.mark.parametrize("kw", [dict(x=np.arange(4.0)), dict(x=np.arange(4.0), y=np.arange(4.0), dx=2.0),
                                dict(dx=0.0), dict(dy=-1.0), dict(x=np.array([0, 1, 2, 4.0]), y=np.arange(4.0)),
                                dict(x=np.arange(5.0), y=np.arange(4.0)), dict(x=np.zeros(4), y=np.arange(4.0))])
def test_calibration_errors(kw):
    with pytest.raises(ValueError):
        C._resolve_steps_2d(shape=(4, 4), x=kw.get("x"), y=kw.get("y"), dx=kw.get("dx", 1.0), dy=kw.get("dy", 1.0))


def test_1d_helpers_match_oracle():
    rng = np.random.default_rng(3)
    a, b = rng.normal(size=77), rng.normal(size=77)
    np.testing.assert_array_equal(F.fft1d(a, dx=0.1)[0], S.fft1d(a, dx=0.1)[0])
    np.testing.assert_array_equal(F.psd1d(a, dx=0.1)[0], S.psd1d(a, dx=0.1)[0])
    np.testing.assert_array_equal(F.ifft1d(F.fft1d(a)[0]), S.ifft1d(S.fft1d(a)[0]))
    np.testing.assert_array_equal(X.xcorr1d(a, b)[0], S.xcorr1d(a, b)[0])
    np.testing.assert_array_equal(X.autocorr1d(a, standardize=True)[0], S.xcorr1d(a, a, standardize=True)[0])
    with pytest.raises(ValueError):
        X.xcorr1d(a, b[:-1])
    with pytest.raises(ValueError):
        X.xcorr1d(a, b, normalize="x")


def test_geometry_matches_oracle():
    assert [G.odd_size(v) for v in (2.2, 3, 4, 10.0)] == [S.odd_size(v) for v in (2.2, 3, 4, 10.0)]
    assert G.odd_size(1, min_size=1) == 1
    with pytest.raises(ValueError):
        G.odd_size(float("nan"))
    for shape, size, c in (((100, 90), (31, 21), None), ((64, 64), (5, 7), (10, 50))):
        assert G.roi_slices(shape, size, center_yx=c) == S.roi_slices(shape, size, center_yx=c)
    assert G.roi_slices((10, 10), (7, 7), center_yx=(1, 9), clip=True) == S.roi_slices((10, 10), (7, 7), center_yx=(1, 9), clip=True)
    for bad in (((10, 10), (4, 5)), ((10, 10), (11, 5)), ((10, 10), (0, 5))):
        with pytest.raises(ValueError):
            G.roi_slices(*bad)
    g1, l1 = G.roi_grid_3x3((200, 300), (21, 31), (40, 50))
    g2, l2 = S.roi_grid_3x3((200, 300), (21, 31), (40, 50))
    assert all(g1[i, j] == g2[i, j] for i in range(3) for j in range(3)) and (l1 == l2).all()
    a = np.arange(12.0).reshape(3, 4)
    np.testing.assert_array_equal(G.pad_to_square(a, fill_value=-1.0), S.pad_to_square(a, fill_value=-1.0))
    np.testing.assert_array_equal(G.pad_to_square(a.T, fill_value=7.0), S.pad_to_square(a.T, fill_value=7.0))
    with pytest.raises(ValueError):
        G.embed_roi(a, out_shape=(8, 8), slices_yx=(slice(0, 2), slice(0, 4)))


def test_shard_bounds_cover_everything():
    from barc4dip_amd.metrics.temporal import shard_bounds

    for total, world in ((8192, 8), (10, 3), (5, 8), (1, 1)):
        spans = [shard_bounds(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_default_chunk_policy():
    from barc4dip_amd import _ffi

    assert _ffi.default_chunk(2048, 2048) == 64
    assert _ffi.default_chunk(4096, 4096) == 16
    assert _ffi.default_chunk(1024, 1024) == 128
    assert 1 <= _ffi.default_chunk(64, 64) <= 128
    assert _ffi.default_chunk(228, 227) == 128 and _ffi.default_chunk(512, 500) >= 32


def test_temporal_sums_layout_is_one_contiguous_buffer():
    """metrics/temporal.py: [count, pad | block 0 | block 1 ...], every block = (sum_x rows, sum_xx rows); the slices
    handed to the collective tile the buffer exactly once and the first one carries the count."""
    import torch

    from barc4dip_amd.metrics.temporal import TemporalSums

    for H, W, k in ((10, 7, 1), (10, 7, 3), (5, 4, 9), (2048, 8, 4)):
        acc = TemporalSums(H, W, torch.device("cpu"), k)
        assert acc.buf.numel() == 2 + 2 * H * W and acc.rows[0][0] == 0 and acc.rows[-1][1] == H
        assert all(a[1] == b[0] for a, b in zip(acc.rows, acc.rows[1:]))
        acc.add_count(5)
        covered = torch.zeros_like(acc.buf)
        for c in range(len(acc.rows)):
            sl = acc.slice_for_reduce(c)
            assert sl.is_contiguous() and sl.data_ptr() >= acc.buf.data_ptr()
            covered[(sl.data_ptr() - acc.buf.data_ptr()) // 8:][:sl.numel()] += 1
            sx, sxx = acc.block(c)
            assert sx.shape == sxx.shape == (acc.rows[c][1] - acc.rows[c][0], W)
            assert sx.data_ptr() % 16 == acc.buf.data_ptr() % 16
        assert torch.all(covered == 1)
        assert float(acc.slice_for_reduce(0)[0]) == 5.0
