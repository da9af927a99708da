"""The own FFT engine in 1-3 dimensions (csrc/fft_pow2.hip: block pass for the last axis, strided passes for the outer
axes) against numpy.fft.fftn, and the convolution / gradient channels that run on it against the oracle and against
the same channels on rocFFT (conv_channel.py:81-109, gradient_channel.py:56-76)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dims(shape):
    return (C.c_size_t * len(shape))(*shape)


def orders(ctx, T, dtype, shape):
    out = []
    for a in range(len(shape)):
        n = shape[a] // 2 if a == len(shape) - 1 else shape[a]
        buf = (C.c_ulonglong * n)()
        ctx._call("tramp_fftn_order", T.device.dtype_code(dtype), len(shape), _dims(shape), a, buf)
        k = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
        assert np.array_equal(np.sort(k), np.arange(n))
        out.append(k)
    return out


# real shapes: one block pass, several rows per tile, the two-pass last axis, one and two strided passes per outer axis
SHAPES = [(16,), (64,), (8192,), (16384,), (20000,), (8, 16), (16, 32), (10, 32), (5, 160), (64, 64), (1, 32),
          (512, 32), (640, 32), (1024, 64), (4096, 32), (250, 1280), (32, 8192), (16, 16384), (4, 8, 32), (2, 5, 64),
          (16, 16, 160), (8, 640, 32), (1024, 4, 32)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_fftn_matches_numpy(ctx, shape, dtype):
    import tramp_amd as T
    dt = T.device.dtype_code(dtype)
    if not T._lib.lib().tramp_fftn_supported(dt, len(shape), _dims(shape)):
        assert dtype == np.float32 and len(shape) > 1 and (shape[-1] // 2) % 16       # f32 lines hold 16 complex
        return
    B = 3
    n = int(np.prod(shape))
    rs = np.random.RandomState(n % 9973)
    x = rs.standard_normal((B,) + shape).astype(dtype)
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    xd = ctx.array(x.reshape(B, n), dtype)
    yd = ctx.empty((B, n), dtype)
    ctx._call("tramp_fftn", dt, len(shape), _dims(shape), B, 0, xd.ptr, yd.ptr)
    cshape = shape[:-1] + (shape[-1] // 2,)
    y = yd.get().view(cdt).reshape((B,) + cshape)
    z = (x[..., 0::2] + 1j * x[..., 1::2]).astype(np.complex128)
    ref = np.fft.fftn(z, axes=tuple(range(1, len(shape) + 1)))
    ks = orders(ctx, T, dtype, shape)
    refp = ref[np.ix_(np.arange(B), *ks)]
    tol = 1e-12 if dtype == np.float64 else 3e-4
    np.testing.assert_allclose(y, refp, rtol=0, atol=tol * np.abs(ref).max())
    zd = ctx.empty((B, n), dtype)
    ctx._call("tramp_fftn", dt, len(shape), _dims(shape), B, 1, yd.ptr, zd.ptr)
    back = zd.get().reshape((B,) + shape) / (n // 2)
    np.testing.assert_allclose(back, x, rtol=0, atol=tol * 10)


def test_fftn_unsupported_shapes(ctx):
    import tramp_amd as T
    dt = T.device.dtype_code(np.float64)
    buf = ctx.empty((1, 4096), np.float64)
    for shape in [(15,), (6, 16), (8, 24), (8, 8), (3, 4, 16), (600000, 16)]:
        assert not T._lib.lib().tramp_fftn_supported(dt, len(shape), _dims(shape))
        with pytest.raises(T._lib.TrampHipError):
            ctx._call("tramp_fftn", dt, len(shape), _dims(shape), 1, 0, buf.ptr, buf.ptr)


def _conv_case(shape, seed):
    rs = np.random.RandomState(seed)
    f = np.zeros(shape)
    idx = tuple(slice(0, min(3, s)) for s in shape)
    f[idx] = rs.standard_normal(f[idx].shape)
    return f, rs


CH_SHAPES = [(64,), (8, 16), (16, 32), (10, 160), (64, 64), (640, 32), (4, 8, 32), (2, 5, 64)]


@pytest.mark.parametrize("shape", CH_SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_conv_channel_on_own_engine(ctx, shape, dtype, monkeypatch):
    """ConvChannel messages, posteriors and x = w * z through the own engine, against the oracle's Fourier-domain
    update and against the same channel built on rocFFT."""
    import tramp_amd as T
    from oracle import tramp_oracle as orc
    if dtype == np.float32 and len(shape) > 1 and (shape[-1] // 2) % 16:
        pytest.skip("f32 lines hold 16 complex")
    f, rs = _conv_case(shape, 7 + len(shape))
    B, n = 3, int(np.prod(shape))
    az, ax = rs.uniform(0.5, 2, B), rs.uniform(0.5, 2, B)
    bz, bx = rs.standard_normal((B,) + shape), rs.standard_normal((B,) + shape)
    res = {}
    for eng in ("own", "rocfft"):
        if eng == "rocfft":
            monkeypatch.setenv("TRAMP_CONV_FFT", "rocfft")
        else:
            monkeypatch.delenv("TRAMP_CONV_FFT", raising=False)
        ch = T.ConvChannel(f, dtype=dtype, ctx=ctx)
        res[eng] = dict(fwd=ch.compute_forward_message(az, bz.astype(dtype), ax, bx.astype(dtype)),
                        bwd=ch.compute_backward_message(az, bz.astype(dtype), ax, bx.astype(dtype)),
                        post=ch.compute_backward_posterior(az, bz.astype(dtype), ax, bx.astype(dtype)),
                        smp=ch.sample(bz[0]))
    tol = 1e-11 if dtype == np.float64 else 3e-4
    ref = orc.ConvOracle(f)
    for b in range(B):
        a_f, b_f = ref.forward_message(az[b], bz[b], ax[b], bx[b])
        a_b, b_b = ref.backward_message(az[b], bz[b], ax[b], bx[b])
        for eng in ("own", "rocfft"):
            r = res[eng]
            np.testing.assert_allclose(r["fwd"][0][b], a_f, rtol=max(tol, 1e-9))
            np.testing.assert_allclose(np.asarray(r["fwd"][1][b]).reshape(shape), b_f, rtol=0, atol=tol * np.abs(b_f).max())
            np.testing.assert_allclose(r["bwd"][0][b], a_b, rtol=max(tol, 1e-9))
            np.testing.assert_allclose(np.asarray(r["bwd"][1][b]).reshape(shape), b_b, rtol=0, atol=tol * np.abs(b_b).max())
    np.testing.assert_allclose(np.asarray(res["own"]["smp"]).reshape(shape), ref.convolve(bz[0]), rtol=0,
                               atol=tol * np.abs(bz[0]).max() * np.abs(f).sum())
    np.testing.assert_allclose(np.asarray(res["own"]["post"][0]), np.asarray(res["rocfft"]["post"][0]), rtol=0,
                               atol=tol * np.abs(np.asarray(res["rocfft"]["post"][0])).max())


@pytest.mark.parametrize("shape", CH_SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gradient_channel_on_own_engine(ctx, shape, dtype, monkeypatch):
    """GradientChannel (d = ndim filters) through the own engine against the oracle and against rocFFT."""
    import tramp_amd as T
    from oracle import tramp_oracle as orc
    if dtype == np.float32 and len(shape) > 1 and (shape[-1] // 2) % 16:
        pytest.skip("f32 lines hold 16 complex")
    rs = np.random.RandomState(11 + len(shape))
    B, d = 2, len(shape)
    az, ax = rs.uniform(0.5, 2, B), rs.uniform(0.5, 2, B)
    bz, bx = rs.standard_normal((B,) + shape), rs.standard_normal((B, d) + shape)
    res = {}
    for eng in ("own", "rocfft"):
        if eng == "rocfft":
            monkeypatch.setenv("TRAMP_CONV_FFT", "rocfft")
        else:
            monkeypatch.delenv("TRAMP_CONV_FFT", raising=False)
        ch = T.GradientChannel(shape, dtype=dtype, ctx=ctx)
        res[eng] = dict(fwd=ch.compute_forward_message(az, bz.astype(dtype), ax, bx.astype(dtype)),
                        bwd=ch.compute_backward_message(az, bz.astype(dtype), ax, bx.astype(dtype)),
                        smp=ch.sample(bz[0]))
    tol = 1e-11 if dtype == np.float64 else 3e-4
    ref = orc.GradientOracle(shape)
    for b in range(B):
        a_f, b_f = ref.forward_message(az[b], bz[b], ax[b], bx[b])
        a_b, b_b = ref.backward_message(az[b], bz[b], ax[b], bx[b])
        for eng in ("own", "rocfft"):
            r = res[eng]
            np.testing.assert_allclose(r["fwd"][0][b], a_f, rtol=max(tol, 1e-9))
            np.testing.assert_allclose(np.asarray(r["fwd"][1][b]).reshape((d,) + shape), b_f, rtol=0,
                                       atol=tol * np.abs(b_f).max())
            np.testing.assert_allclose(r["bwd"][0][b], a_b, rtol=max(tol, 1e-9))
            np.testing.assert_allclose(np.asarray(r["bwd"][1][b]).reshape(shape), b_b, rtol=0, atol=tol * np.abs(b_b).max())
    np.testing.assert_allclose(np.asarray(res["own"]["smp"]).reshape((d,) + shape), ref.convolve(bz[0]), rtol=0,
                               atol=tol * 4 * np.abs(bz[0]).max())
