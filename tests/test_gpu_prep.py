"""GPU tier: flat_field_correction through the C ABI against the oracle and the reference's golden outputs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "prep.npz"))


@pytest.fixture(scope="module")
def prep():
    from barc4dip_amd import preprocessing
    return preprocessing


def _cases():
    from test_prep_oracle import cases
    return cases()


@pytest.mark.parametrize("name", ["default", "none", "repair", "eps50", "single", "flat_only", "dark_only", "neither", "f32_in"])
def test_flat_field_bit_exact(prep, name):
    """float32 arithmetic in the reference's order: identical bits (normalize.py:12-145)."""
    img, kw = _cases()[name]
    out = prep.flat_field_correction(img, **kw)
    assert isinstance(out, np.ndarray) and out.dtype == np.float32 and out.shape == G[name].shape
    assert np.array_equal(out, G[name], equal_nan=True)


def test_flat_mean_scale(prep):
    """flat_mean: NumPy sums the valid denominators pairwise in float32, the kernel in float64: <= 1 ulp of the scale
    factor, i.e. 2^-23 relative plus one rounding on the output."""
    img, kw = _cases()["mean"]
    out = prep.flat_field_correction(img, **kw)
    np.testing.assert_allclose(out, G["mean"], rtol=3e-7, atol=0)
    assert np.array_equal(out == 0, G["mean"] == 0)


def test_flat_field_large_vs_oracle(prep):
    from barc4dip_amd import synth
    from oracle import preprocess_np as P

    rng = np.random.default_rng(11)
    n = 512
    imgs = np.stack([synth.speckle_frame(n, 70 + i) for i in range(3)]).astype(np.uint16)
    gain = 1.0 + 0.2 * np.sin(np.arange(n) / 13.0)[None, :] * np.cos(np.arange(n) / 17.0)[:, None]
    flats = rng.poisson(3000 * gain[None] + 90, size=(4, n, n)).astype(np.uint16)
    darks = rng.poisson(90, size=(6, n, n)).astype(np.uint16)
    flats[:, rng.integers(0, n, 40), rng.integers(0, n, 40)] = 0
    flats[:, 0, :7] = 0
    flats[:, -1, -3:] = 0
    for kw in (dict(), dict(bad_pixel_removal=True), dict(scale="none", eps=10.0, bad_pixel_removal=True)):
        got = prep.flat_field_correction(imgs, flats=flats, darks=darks, **kw)
        want = P.flat_field_correction(imgs, flats=flats, darks=darks, **kw)
        assert np.array_equal(got, want, equal_nan=True), kw


def test_flat_field_errors(prep):
    im = G["imgs"]
    with pytest.raises(ValueError):
        prep.flat_field_correction(im, flats=G["flats"], scale="median")
    with pytest.raises(ValueError):
        prep.flat_field_correction(im[0, 0], flats=G["flats"])
    with pytest.raises(ValueError):
        prep.flat_field_correction(im, flats=G["flats"][None])


def test_nan_semantics(prep):
    """NaN propagation like NumPy's: a NaN flat pixel makes the median / mean scale NaN (whole output NaN), np.clip keeps
    NaN in deconvolve_psf (a NaN pixel spreads over the whole frame through the FFT)."""
    from barc4dip_amd import synth
    from oracle import preprocess_np as P
    from oracle import wiener_np as W

    img = synth.speckle_frame(200, 5)[:150, :180]
    rng = np.random.default_rng(1)
    flats = rng.poisson(2000, size=(3, 150, 180)).astype(np.float32)
    darks = rng.poisson(100, size=(150, 180)).astype(np.float32)
    flats[:, 3, 4] = np.nan
    for kw in (dict(), dict(bad_pixel_removal=True), dict(scale="none"), dict(scale="flat_mean")):
        got = prep.flat_field_correction(img, flats=flats, darks=darks, **kw)
        want = P.flat_field_correction(img, flats=flats, darks=darks, **kw)
        assert np.array_equal(np.isnan(got), np.isnan(want)), kw
        assert np.array_equal(got, want, equal_nan=True) or kw.get("scale") == "flat_mean"
    nanimg = img.copy()
    nanimg[5, 7] = np.nan
    got, want = prep.deconvolve_psf(nanimg, sigma=1.0), W.deconvolve_psf(nanimg, sigma=1.0)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got).all()
    z = np.zeros_like(img)
    assert np.array_equal(prep.deconvolve_psf(z, sigma=1.0), W.deconvolve_psf(z, sigma=1.0))


def test_to_f32_every_dtype_aligned_and_not(prep):
    """b4d_to_f32 (images.astype(np.float32), normalize.py:79 / the ingest path): every detector word type, lengths with a tail
    behind the 16-byte vector part, and a source view that starts off a 16-byte boundary (scalar kernel)."""
    import ctypes as C

    import torch

    from barc4dip_amd import _ffi

    lib = _ffi.lib()
    rng = np.random.default_rng(0)
    kinds = [(np.uint8, 0), (np.uint16, 1), (np.int16, 2), (np.int32, 3), (np.uint32, 4), (np.float32, 5), (np.float64, 6)]
    for dt, code in kinds:
        for n, off in ((1, 0), (37, 0), (4096 + 5, 0), (100003, 0), (4096 + 5, 1)):
            if np.issubdtype(dt, np.integer):
                info = np.iinfo(dt)
                host = rng.integers(info.min, info.max, size=n + off, endpoint=True, dtype=dt)
            else:
                host = (rng.normal(size=n + off) * 1e3).astype(dt)
            raw = torch.from_numpy(host.view(np.uint8).copy()).cuda()
            src_ptr = raw.data_ptr() + off * host.itemsize
            out = torch.full((n,), -1.0, dtype=torch.float32, device="cuda")
            _ffi.check(lib.b4d_to_f32(C.c_void_p(src_ptr), code, n, C.c_void_p(out.data_ptr()), None))
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), host[off:].astype(np.float32)), (dt, n, off)
    assert lib.b4d_to_f32(C.c_void_p(raw.data_ptr()), 9, 4, C.c_void_p(out.data_ptr()), None) != 0


@pytest.mark.parametrize("dtype", ["uint8", "uint16", "int16", "int32", "uint32", "float32", "float64"])
def test_large_host_arrays_go_up_in_their_native_dtype(prep, dtype):
    """_device.to_device_f32 (every entry point's way in): host arrays of 32 MiB and more are staged through page-locked blocks in
    their native dtype and converted on the device -- bit-identical to ndarray.astype(float32), for sizes that do not divide into
    the blocks, values at the ends of the integer ranges, NaN / Inf, and the small-array route next to it."""
    import torch

    from barc4dip_amd import _device as D

    rng = np.random.default_rng(11)
    dt = np.dtype(dtype)
    n_items = (40 << 20) // dt.itemsize + 12345          # > one block, not a multiple of anything
    if dt.kind in "ui":
        info = np.iinfo(dt)
        a = rng.integers(info.min, info.max, size=n_items, dtype=dt, endpoint=True)
        a[:4] = (info.max, info.min, info.max - 1, 0)
    else:
        a = (rng.standard_normal(n_items) * 1e4).astype(dt)
        a[:6] = (np.nan, np.inf, -np.inf, 0.0, -0.0, 1e-40)
    shape = (3, 1, n_items // 3) if n_items % 3 == 0 else (1, 1, n_items)
    a = a.reshape(shape)
    assert a.nbytes >= D._UPLOAD_MIN_BYTES
    t, was_tensor, src = D.to_device_f32(a, ndim=(3,))
    assert not was_tensor and t.dtype == torch.float32 and tuple(t.shape) == shape
    assert src == (np.float32 if dtype == "float32" else np.float64)
    np.testing.assert_array_equal(t.cpu().numpy().view(np.uint32), a.astype(np.float32).view(np.uint32))
    small = a.reshape(-1)[:100003].reshape(1, 1, -1)
    np.testing.assert_array_equal(D.to_device_f32(small, ndim=(3,))[0].cpu().numpy().view(np.uint32),
                                  small.astype(np.float32).view(np.uint32))
    # a non-contiguous view of a large array takes the ordinary route
    v = a[:, :, ::2]
    np.testing.assert_array_equal(D.to_device_f32(v, ndim=(3,))[0].cpu().numpy().view(np.uint32), v.astype(np.float32).view(np.uint32))


def test_large_results_come_down_through_page_locked_blocks(prep):
    """_device.to_host beyond its whole-result page-locking limit (_download_staged): device-side conversion block by block, host
    threads copying out of two page-locked blocks -- equal to .cpu().numpy().astype(...), block count odd / even, ragged tail."""
    import torch

    from barc4dip_amd import _device as D

    t = torch.randn(3, 7, 1234567, device="cuda") * 1e3          # 25.9 M elements: 4 blocks as float32, 7 as float64
    t[0, 0, :3] = torch.tensor([float("nan"), float("inf"), -0.0], device="cuda")
    ref = t.cpu().numpy()
    for want in (torch.float64, torch.float32):
        got = D._download_staged(t, want)
        assert got.shape == tuple(t.shape) and got.dtype == (np.float64 if want == torch.float64 else np.float32)
        np.testing.assert_array_equal(got, ref.astype(got.dtype))
    old = D._PINNED_MAX_BYTES
    try:
        D._PINNED_MAX_BYTES = 16 << 20                           # route the public entry through it
        np.testing.assert_array_equal(D.to_host(t, np.float64), ref.astype(np.float64))
    finally:
        D._PINNED_MAX_BYTES = old
