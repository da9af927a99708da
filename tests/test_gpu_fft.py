"""The hand-written two-pass mixed-radix (2^a 5^b) FFT (csrc/fft_pow2.hip) against numpy.fft."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def kfreq(ctx, h):
    out = (C.c_ulonglong * h)()
    ctx._call("tramp_fft_order", h, h, out)
    return np.frombuffer(out, dtype=np.uint64).astype(np.int64)


LENGTHS = [1 << 10, 1 << 11, 1 << 12, 1 << 13, 1 << 15, 1 << 16, 1 << 18, 1 << 19, 1 << 20, 5120, 10000, 12800,
           50000, 80000, 500000, 1280000]


@pytest.mark.parametrize("h", LENGTHS)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fft_matches_numpy(ctx, h, dtype):
    import tramp_amd as T
    if h >= 1 << 20 and dtype == np.float32:
        pytest.skip("covered by the f64 case")
    B = 3 if h <= 1 << 19 else 1
    rs = np.random.RandomState(h % 9973)
    x = (rs.standard_normal((B, h)) + 1j * rs.standard_normal((B, h)))
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    xi = np.ascontiguousarray(x.astype(cdt)).view(dtype).reshape(B, 2 * h)
    xd = ctx.array(xi, dtype)
    yd = ctx.empty((B, 2 * h), dtype)
    ctx._call("tramp_fft_len", T.device.dtype_code(dtype), h, B, 0, xd.ptr, yd.ptr)
    y = yd.get().view(cdt).reshape(B, h)
    k = kfreq(ctx, h)
    assert np.array_equal(np.sort(k), np.arange(h))                      # a permutation
    ref = np.fft.fft(x.astype(cdt).astype(np.complex128), axis=1)
    tol = 1e-12 if dtype == np.float64 else 2e-4
    np.testing.assert_allclose(y, ref[:, k], rtol=0, atol=tol * np.abs(ref).max())
    # inverse: permuted spectrum -> natural order, unnormalised
    zd = ctx.empty((B, 2 * h), dtype)
    ctx._call("tramp_fft_len", T.device.dtype_code(dtype), h, B, 1, yd.ptr, zd.ptr)
    z = zd.get().view(cdt).reshape(B, h) / h
    np.testing.assert_allclose(z, x.astype(cdt), rtol=0, atol=tol * 10)


def test_fft_unsupported_lengths(ctx):
    import tramp_amd as T
    buf = ctx.empty((1, 64), np.float64)
    for h in (32, 1000, 3 * 4096, 7 * 1024, 1 << 22):     # too short, LB not a multiple of 16, factors 3 / 7, too long
        with pytest.raises(T._lib.TrampHipError):
            ctx._call("tramp_fft_len", T.device.dtype_code(np.float64), h, 1, 0, buf.ptr, buf.ptr)
