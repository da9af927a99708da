"""The hand-written two-pass power-of-two FFT (csrc/fft_pow2.hip) against numpy.fft."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def kfreq(ctx, q):
    n = 1 << q
    out = (C.c_ulonglong * n)()
    ctx._call("tramp_fft_pow2_order", q, n, out)
    return np.frombuffer(out, dtype=np.uint64).astype(np.int64)


@pytest.mark.parametrize("q", [12, 13, 14, 15, 16, 17, 18, 19, 20, 21])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fft_pow2_matches_numpy(ctx, q, dtype):
    import tramp_amd as T
    if q >= 20 and dtype == np.float32:
        pytest.skip("covered by the f64 case")
    h, B = 1 << q, (3 if q <= 19 else 1)
    rs = np.random.RandomState(q)
    x = (rs.standard_normal((B, h)) + 1j * rs.standard_normal((B, h)))
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    xi = np.ascontiguousarray(x.astype(cdt)).view(dtype).reshape(B, 2 * h)
    xd = ctx.array(xi, dtype)
    yd = ctx.empty((B, 2 * h), dtype)
    ctx._call("tramp_fft_pow2", T.device.dtype_code(dtype), q, B, 0, xd.ptr, yd.ptr)
    y = yd.get().view(cdt).reshape(B, h)
    k = kfreq(ctx, q)
    assert np.array_equal(np.sort(k), np.arange(h))                      # a permutation
    ref = np.fft.fft(x.astype(cdt).astype(np.complex128), axis=1)
    tol = 1e-12 if dtype == np.float64 else 2e-4
    np.testing.assert_allclose(y, ref[:, k], rtol=0, atol=tol * np.abs(ref).max())
    # inverse: permuted spectrum -> natural order, unnormalised
    zd = ctx.empty((B, 2 * h), dtype)
    ctx._call("tramp_fft_pow2", T.device.dtype_code(dtype), q, B, 1, yd.ptr, zd.ptr)
    z = zd.get().view(cdt).reshape(B, h) / h
    np.testing.assert_allclose(z, x.astype(cdt), rtol=0, atol=tol * 10)


def test_fft_pow2_unsupported_lengths(ctx):
    import tramp_amd as T
    buf = ctx.empty((1, 64), np.float64)
    for q in (5, 11, 22):
        with pytest.raises(T._lib.TrampHipError):
            ctx._call("tramp_fft_pow2", T.device.dtype_code(np.float64), q, 1, 0, buf.ptr, buf.ptr)
