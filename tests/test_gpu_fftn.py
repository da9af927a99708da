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
    if not T._lib.load().tramp_fftn_supported(dt, len(shape), _dims(shape)):
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
        assert not T._lib.load().tramp_fftn_supported(dt, len(shape), _dims(shape))
        with pytest.raises(T._lib.TrampHipError):
            ctx._call("tramp_fftn", dt, len(shape), _dims(shape), 1, 0, buf.ptr, buf.ptr)


def _conv_case(shape, seed):
    rs = np.random.RandomState(seed)
    f = np.zeros(shape)
    idx = tuple(slice(0, min(3, s)) for s in shape)
    f[idx] = rs.standard_normal(f[idx].shape)
    return f, rs


CH_SHAPES = [(64,), (4096,), (8192,), (16384,), (8, 16), (16, 32), (10, 160), (64, 64), (640, 32), (4, 8, 32), (2, 5, 64)]


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
            monkeypatch.setenv("TRAMP_CONV_FFT", "own")
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
            monkeypatch.setenv("TRAMP_CONV_FFT", "own")
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


DFT_SHAPES = [(64,), (1000,), (8192,), (16, 32), (10, 160), (640, 32), (4, 8, 16), (15,), (6, 8)]


@pytest.mark.parametrize("shape", DFT_SHAPES + [(640, 32), (1024, 4, 16)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("real", [True, False])
@pytest.mark.parametrize("engine", ["own", "rocfft"])
def test_dft_channel_real_and_complex_input(ctx, shape, dtype, real, engine, monkeypatch):
    """DFTChannel messages (dft_channel.py:47-67) on rocFFT (TRAMP_DFT_FFT=rocfft) and on the own engine's natural-order
    transforms (the default since round 5; shapes it does not take - 15, 6x8, a last axis beyond one block - stay on rocFFT), for real z and
    for complex z (real=False: planar [2, ...] arrays on both sides)."""
    import tramp_amd as T
    if engine == "own":
        monkeypatch.setenv("TRAMP_DFT_FFT", "own")
    else:
        monkeypatch.setenv("TRAMP_DFT_FFT", "rocfft")
    rs = np.random.RandomState(len(shape) + int(real))
    B, n = 3, int(np.prod(shape))
    ch = T.DFTChannel(real=real, shape=shape, dtype=dtype, ctx=ctx)
    az, ax = rs.uniform(0.5, 2, B), rs.uniform(0.5, 2, B)
    zc = rs.standard_normal((B,) + shape) + (0 if real else 1j * rs.standard_normal((B,) + shape))
    xc = rs.standard_normal((B,) + shape) + 1j * rs.standard_normal((B,) + shape)
    planar = lambda c: np.stack([c.real, c.imag], axis=1)          # noqa: E731  [B, 2, ...]
    bz = zc.real if real else planar(zc)
    bx = planar(xc)
    axes = tuple(range(1, len(shape) + 1))
    tol = 1e-12 if dtype == np.float64 else 3e-5
    a_f, b_f = ch.compute_forward_message(az, bz.astype(dtype), ax, bx.astype(dtype))
    ref_f = planar(np.fft.fftn(zc, axes=axes, norm="ortho"))
    np.testing.assert_allclose(a_f, az, rtol=1e-15)
    np.testing.assert_allclose(np.asarray(b_f).reshape(ref_f.shape), ref_f, rtol=0, atol=tol * np.abs(ref_f).max() * np.sqrt(n))
    a_b, b_b = ch.compute_backward_message(az, bz.astype(dtype), ax, bx.astype(dtype))
    inv = np.fft.ifftn(xc, axes=axes, norm="ortho")
    ref_b = inv.real if real else planar(inv)
    np.testing.assert_allclose(a_b, ax, rtol=1e-15)
    np.testing.assert_allclose(np.asarray(b_b).reshape(ref_b.shape), ref_b, rtol=0, atol=tol * np.abs(inv).max() * np.sqrt(n))
    # one instance, host vectors in the reference's shapes
    a1, b1 = ch.compute_forward_message(1.5, bz[0].astype(dtype), 0.5, bx[0].astype(dtype))
    assert np.asarray(b1).shape == (2,) + shape and a1 == 1.5
    a2, b2 = ch.compute_backward_message(1.5, bz[0].astype(dtype), 0.5, bx[0].astype(dtype))
    assert np.asarray(b2).shape == (shape if real else (2,) + shape) and a2 == 0.5
    np.testing.assert_allclose(b2, ref_b[0], rtol=0, atol=tol * np.abs(inv).max() * np.sqrt(n))


def _load_complex_goldens():
    import os
    from tests.conftest import GOLDEN
    with np.load(os.path.join(GOLDEN, "conv_complex.npz"), allow_pickle=True) as f:
        return list(f["data"]), list(f["dft"])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_complex_conv_channel_matches_reference_goldens(ctx, dtype, monkeypatch):
    """ConvChannel(real=False) against the reference's own outputs (conv_channel.py:25-32, 81-109): posteriors, messages,
    x = w * z, n_eff and second moment; shapes on the own engine (16, 64, 8x16, 16x32, 4x4x16) and on rocFFT (15, 6x8, 3x4x5)."""
    import tramp_amd as T
    monkeypatch.setenv("TRAMP_CONV_FFT", "own")          # small shapes: the default would be rocFFT
    conv, _ = _load_complex_goldens()
    tol = 1e-11 if dtype == np.float64 else 3e-4
    chans = {}
    for c in conv:
        key = tuple(c["shape"])
        if key not in chans:
            chans[key] = T.ConvChannel(c["filter"], real=False, dtype=dtype, ctx=ctx)
        ch = chans[key]
        az, ax, bz, bx = c["az"], c["ax"], c["bz"].astype(dtype), c["bx"].astype(dtype)
        np.testing.assert_allclose(ch.compute_n_eff(az, ax), c["n_eff"], rtol=max(tol, 1e-9))
        np.testing.assert_allclose(ch.second_moment(1.7), c["tau"], rtol=max(tol, 1e-9))
        rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
        rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
        assert np.asarray(rz).shape == (2,) + key
        np.testing.assert_allclose(rz, c["rz"], rtol=0, atol=tol * np.abs(c["rz"]).max())
        np.testing.assert_allclose(rx, c["rx"], rtol=0, atol=tol * np.abs(c["rx"]).max())
        np.testing.assert_allclose([vz, vx], [c["vz"], c["vx"]], rtol=max(tol, 1e-9))
        a_b, b_b = ch.compute_backward_message(az, bz, ax, bx)
        a_f, b_f = ch.compute_forward_message(az, bz, ax, bx)
        np.testing.assert_allclose(a_b, c["a_bwd"], rtol=max(tol, 1e-9), atol=max(tol, 1e-9) / c["vz"])
        np.testing.assert_allclose(a_f, c["a_fwd"], rtol=max(tol, 1e-9), atol=max(tol, 1e-9) / c["vx"])
        sb = np.abs(c["rz"]).max() / c["vz"] + np.abs(bz).max()
        sf = np.abs(c["rx"]).max() / c["vx"] + np.abs(bx).max()
        np.testing.assert_allclose(b_b, c["b_bwd"], rtol=0, atol=tol * sb)
        np.testing.assert_allclose(b_f, c["b_fwd"], rtol=0, atol=tol * sf)
        np.testing.assert_allclose(ch.sample(bz), c["sample"], rtol=0, atol=tol * np.abs(c["sample"]).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("engine", ["own", "rocfft"])
def test_complex_dft_channel_matches_reference_goldens(ctx, dtype, engine, monkeypatch):
    import tramp_amd as T
    if engine == "own":
        monkeypatch.setenv("TRAMP_DFT_FFT", "own")
    else:
        monkeypatch.setenv("TRAMP_DFT_FFT", "rocfft")
    _, dft = _load_complex_goldens()
    tol = 1e-12 if dtype == np.float64 else 3e-5
    for c in dft:
        shape = tuple(c["shape"])
        ch = T.DFTChannel(real=False, shape=shape, dtype=dtype, ctx=ctx)
        a_f, b_f = ch.compute_forward_message(1.3, c["bz"].astype(dtype), 0.7, c["bx"].astype(dtype))
        a_b, b_b = ch.compute_backward_message(1.3, c["bz"].astype(dtype), 0.7, c["bx"].astype(dtype))
        assert (a_f, a_b) == (c["a_fwd"], c["a_bwd"])
        sc = np.sqrt(np.prod(shape))
        np.testing.assert_allclose(b_f, c["b_fwd"], rtol=0, atol=tol * np.abs(c["b_fwd"]).max() * sc)
        np.testing.assert_allclose(b_b, c["b_bwd"], rtol=0, atol=tol * np.abs(c["b_bwd"]).max() * sc)
        np.testing.assert_allclose(ch.sample(c["bz"].astype(dtype)), c["sample"], rtol=0,
                                   atol=tol * np.abs(c["sample"]).max() * sc)


@pytest.mark.parametrize("shape", [(4096,), (64, 64), (640, 32), (8, 16, 32)], ids=lambda s: "x".join(map(str, s)))
def test_complex_conv_channel_larger_shapes_and_damping(ctx, shape, monkeypatch):
    "Own engine against the oracle and against rocFFT at sizes with strided passes; an ensemble with damping."
    import tramp_amd as T
    from oracle import tramp_oracle as orc
    rs = np.random.RandomState(len(shape))
    f = np.zeros(shape, dtype=complex)
    idx = tuple(slice(0, 3) for _ in shape)
    f[idx] = rs.standard_normal(f[idx].shape) + 1j * rs.standard_normal(f[idx].shape)
    B = 3
    az, ax = rs.uniform(0.5, 2, B), rs.uniform(0.5, 2, B)
    bz, bx = rs.standard_normal((B, 2) + shape), rs.standard_normal((B, 2) + shape)
    o = orc.ComplexConvOracle(f)
    out = {}
    for eng in ("own", "rocfft"):
        if eng == "rocfft":
            monkeypatch.setenv("TRAMP_CONV_FFT", "rocfft")
        else:
            monkeypatch.setenv("TRAMP_CONV_FFT", "own")
        ch = T.ConvChannel(f, real=False, ctx=ctx)
        out[eng] = (ch.compute_forward_message(az, bz, ax, bx), ch.compute_backward_message(az, bz, ax, bx))
        # damping through the device entry (old message = the incoming one)
        old_a, old_b = ctx.array(np.full(B, 0.9)), ctx.array(bx.reshape(B, -1).copy())
        a_d, b_d = ch.compute_forward_message(ctx.array(az), ctx.array(bz.reshape(B, -1)), ctx.array(ax),
                                              ctx.array(bx.reshape(B, -1)), damping=0.25, old_a=old_a, old_b=old_b,
                                              out_a=ctx.empty((B,), np.float64), out_b=ctx.empty((B, 2 * int(np.prod(shape)))))
        out[eng + "_damped"] = (a_d.get(), b_d.get())
    for b in range(B):
        ra, rb = o.forward_message(az[b], bz[b], ax[b], bx[b])
        qa, qb = o.backward_message(az[b], bz[b], ax[b], bx[b])
        for eng in ("own", "rocfft"):
            (a_f, b_f), (a_b, b_b) = out[eng]
            np.testing.assert_allclose(a_f[b], ra, rtol=1e-9)
            np.testing.assert_allclose(b_f[b], rb, rtol=0, atol=1e-11 * np.abs(rb).max())
            np.testing.assert_allclose(a_b[b], qa, rtol=1e-9)
            np.testing.assert_allclose(b_b[b], qb, rtol=0, atol=1e-11 * np.abs(qb).max())
            a_d, b_d = out[eng + "_damped"]
            np.testing.assert_allclose(a_d[b], 0.25 * 0.9 + 0.75 * ra, rtol=1e-9)
            np.testing.assert_allclose(b_d[b].reshape(rb.shape), 0.25 * bx[b] + 0.75 * rb, rtol=0, atol=1e-11 * np.abs(rb).max())


def test_gradient_channel_real_false_returns_complex(ctx):
    "real=False only drops np.real() in the reference (gradient_channel.py:32-33, 73-74, 82-83): complex results, same values"
    import tramp_amd as T
    rs = np.random.RandomState(1)
    shape = (8, 16)
    a, b = T.GradientChannel(shape, ctx=ctx), T.GradientChannel(shape, real=False, ctx=ctx)
    bz, bx = rs.standard_normal(shape), rs.standard_normal((2,) + shape)
    ra, va = a.compute_forward_posterior(1.2, bz, 0.8, bx)
    rb, vb = b.compute_forward_posterior(1.2, bz, 0.8, bx)
    assert np.iscomplexobj(rb) and not np.iscomplexobj(ra) and va == vb
    assert np.array_equal(rb.real, ra) and not rb.imag.any()
    assert np.iscomplexobj(b.sample(bz)) and np.array_equal(b.sample(bz).real, a.sample(bz))


def _piecewise_image(shape, rs, rho=0.05):
    img = np.cumsum(np.cumsum(rs.standard_normal(shape) * rs.binomial(1, rho, shape), axis=0), axis=1)
    return img - img.mean()


@pytest.mark.parametrize("shape", [(32, 64), (8, 16, 32)], ids=lambda s: "x".join(map(str, s)))
def test_tree_ep_on_own_engine_matches_oracle_and_rocfft(ctx, shape, monkeypatch):
    """The sparse-gradient tree model (examples/sparse/plot_sparse_gradient.py) on images / volumes whose transforms
    run on the own N-D engine: same trajectory end point as the CPU oracle's tree EP and as the same model on rocFFT;
    graph replay identical to the eager loop."""
    import tramp_amd as T
    from oracle import tramp_oracle as orc
    from tests.test_gpu_tree import TREE_EDGES, build_tree
    rs = np.random.RandomState(7)
    d, n, rho, nv = len(shape), int(np.prod(shape)), 0.05, 1e-2
    x = _piecewise_image(shape, rs)
    y = x + np.sqrt(nv) * rs.standard_normal(shape)
    factors = dict(GaussianPrior=orc.GaussPriorFactor(shape), GaussianLikelihood=orc.GaussLikFactor(y, nv),
                   GradientChannel=orc.GradientFactor(shape), GaussBernoulliPrior=orc.GBPriorFactor((d,) + shape, rho=rho))
    want = orc.ep_tree(factors, ["x", "z"], TREE_EDGES, dict(x=shape, z=(d,) + shape), max_iter=12, damping=0.1, tol=0.0)
    res = {}
    for eng, graph in (("own", False), ("own", True), ("rocfft", False)):
        if eng == "rocfft":
            monkeypatch.setenv("TRAMP_CONV_FFT", "rocfft")
        else:
            monkeypatch.setenv("TRAMP_CONV_FFT", "own")
        model = build_tree(T, ctx, dict(shape=shape, noise_var=nv, rho=rho), y=y)
        ep = T.ExpectationPropagation(model, ctx=ctx).iterate(max_iter=12, damping=0.1, callback=T.PassCallback(),
                                                              graph=graph)
        res[(eng, graph)] = (np.asarray(ep.get_variable_data("x")["r"]).reshape(shape),
                             np.asarray(ep.get_variable_data("z")["r"]).reshape((d,) + shape))
    assert np.array_equal(res[("own", False)][0], res[("own", True)][0])
    assert np.array_equal(res[("own", False)][1], res[("own", True)][1])
    for key in (("own", False), ("rocfft", False)):
        np.testing.assert_allclose(res[key][0], want["r"]["x"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(res[key][1], want["r"]["z"], rtol=1e-7, atol=1e-8)


def test_conv2d_ep_graph_replay_on_own_engine(ctx, monkeypatch):
    "A 2-D deconvolution EP on the N-D plans: iterate(graph=True) gives the bits of the eager loop"
    import tramp_amd as T
    monkeypatch.setenv("TRAMP_CONV_FFT", "own")
    rs = np.random.RandomState(3)
    shape, B, rho, nv = (32, 64), 2, 0.1, 1e-3
    n = shape[0] * shape[1]
    f = rs.standard_normal(shape) / np.sqrt(n)
    X = rs.standard_normal((B, n)) * rs.binomial(1, rho, (B, n))
    outs = []
    for graph in (False, True):
        ch = T.ConvChannel(f, ctx=ctx)
        Y = ch.convolve(X.reshape((B,) + shape)).reshape(B, n) + np.sqrt(nv) * np.random.RandomState(4).standard_normal((B, n))
        model = (T.GaussBernoulliPrior(size=shape, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ ch @ T.SISOVariable(id="z")
                 @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
        ep = T.ExpectationPropagation(model, ctx=ctx).iterate(max_iter=10, damping=0.1, callback=T.PassCallback(), graph=graph)
        outs.append(np.asarray(ep.get_variables_data()["x"]["r"]).copy())
    assert np.array_equal(outs[0], outs[1])
    assert np.mean((outs[0].reshape(B, n) - X) ** 2) < 0.2 * np.mean(X ** 2)


@pytest.mark.parametrize("shape", [(16, 32), (12, 10)], ids=lambda s: "x".join(map(str, s)))
def test_gradient_channel_shares_transforms_between_directions(ctx, shape, monkeypatch):
    """Device-resident messages that were not rewritten since the previous call are not transformed again
    (TRAMP_REUSE_T / _U, as the convolution channel): same results as a fresh channel; a rewritten buffer (version bump)
    is transformed anew.  (16, 32) runs on the own engine, (12, 10) on rocFFT."""
    import tramp_amd as T
    monkeypatch.setenv("TRAMP_CONV_FFT", "own")
    rs = np.random.RandomState(2)
    B, d, n = 2, len(shape), int(np.prod(shape))
    az, ax = ctx.array(rs.uniform(0.5, 2, B)), ctx.array(rs.uniform(0.5, 2, B))
    bz, bx = ctx.array(rs.standard_normal((B, n))), ctx.array(rs.standard_normal((B, d * n)))
    bz.version = bx.version = 0
    ch = T.GradientChannel(shape, ctx=ctx)

    def both(channel):
        f = channel.compute_forward_message(az, bz, ax, bx)
        b = channel.compute_backward_message(az, bz, ax, bx)       # second call: both transforms reused on `ch`
        return f[0].get(), f[1].get(), b[0].get(), b[1].get()

    got = both(ch)
    want = both(T.GradientChannel(shape, ctx=ctx))
    bz.version = bx.version = None                                  # unversioned: never reused
    plain = both(T.GradientChannel(shape, ctx=ctx))
    for g, w, p in zip(got, want, plain):
        assert np.array_equal(g, w) and np.array_equal(g, p)
    # rewrite bz in place: the version bump invalidates its transform, bx's stays
    bz.version = bx.version = 0
    both(ch)
    bz.set(rs.standard_normal((B, n)))
    bz.version += 1
    again = both(ch)
    fresh = both(T.GradientChannel(shape, ctx=ctx))
    for g, w in zip(again, fresh):
        assert np.array_equal(g, w)


@pytest.mark.parametrize("shape,dtype", [((4096, 4096), np.float64), ((4096, 4096), np.float32), ((256, 256, 512), np.float32)],
                         ids=["4096x4096-f64", "4096x4096-f32", "256x256x512-f32"])
def test_fftn_full_size_properties(ctx, shape, dtype):
    """At image scale (three passes per transform; the sizes of tools/bench_tree.py) through size-independent properties:
    the inverse of the forward transform is h times the input; Parseval; and the transform of a plane wave is one peak at
    the position the per-axis orders predict."""
    import tramp_amd as T
    dt = T.device.dtype_code(dtype)
    n = int(np.prod(shape))
    B = 2
    x = ctx.randn((B, n), dtype, seed=5)
    y, z = ctx.empty((B, n), dtype), ctx.empty((B, n), dtype)
    ctx._call("tramp_fftn", dt, len(shape), _dims(shape), B, 0, x.ptr, y.ptr)
    xh = x.get()
    yh = y.get()
    tol = 1e-12 if dtype == np.float64 else 2e-4
    e_in = float(np.sum(xh.astype(np.float64) ** 2))
    e_out = float(np.sum(yh.astype(np.float64) ** 2)) / (n // 2)
    assert abs(e_out - e_in) < (1e-11 if dtype == np.float64 else 1e-4) * e_in
    ctx._call("tramp_fftn", dt, len(shape), _dims(shape), B, 1, y.ptr, z.ptr)
    zh = z.get() / (n // 2)
    assert np.abs(zh - xh).max() < tol * 10 * np.abs(xh).max()
    # plane wave of the packed array with frequency (k0, .., kl): exp(2 pi i (k . j / dims))
    cshape = shape[:-1] + (shape[-1] // 2,)
    ks = tuple(int(c // 3 + 1) for c in cshape)
    phase = sum(np.arange(c).reshape([-1 if a == i else 1 for a in range(len(cshape))]) * (k / c)
                for i, (c, k) in enumerate(zip(cshape, ks)))
    w = np.exp(2j * np.pi * phase)
    wave = np.empty(shape, dtype=dtype)
    wave[..., 0::2], wave[..., 1::2] = w.real, w.imag
    wd = ctx.array(wave.reshape(1, n), dtype)
    od = ctx.empty((1, n), dtype)
    ctx._call("tramp_fftn", dt, len(shape), _dims(shape), 1, 0, wd.ptr, od.ptr)
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    spec = od.get().view(cdt).reshape(cshape)
    orders = [np.asarray(o) for o in globals()["orders"](ctx, T, dtype, shape)]
    pos = tuple(int(np.where(o == k)[0][0]) for o, k in zip(orders, ks))
    peak = spec[pos]
    assert abs(peak - n // 2) < (1e-9 if dtype == np.float64 else 1e-3) * (n // 2)
    spec[pos] = 0
    assert np.abs(spec).max() < (1e-9 if dtype == np.float64 else 2e-3) * (n // 2)


@pytest.mark.parametrize("kind", ["conv", "grad"])
def test_nd_channels_extreme_precisions(ctx, kind, monkeypatch):
    "The N-D spectral solve at precisions spanning sixteen decades (the clip range of base.py:238-255), against the oracle"
    import tramp_amd as T
    from oracle import tramp_oracle as orc
    monkeypatch.setenv("TRAMP_CONV_FFT", "own")
    shape = (16, 32)
    rs = np.random.RandomState(8)
    if kind == "conv":
        f = np.zeros(shape)
        f[:3, :3] = rs.standard_normal((3, 3))
        ch, o, xs = T.ConvChannel(f, ctx=ctx), orc.ConvOracle(f), shape
    else:
        ch, o, xs = T.GradientChannel(shape, ctx=ctx), orc.GradientOracle(shape), (2,) + shape
    bz, bx = rs.standard_normal(shape), rs.standard_normal(xs)
    for az, ax in [(1e-8, 1.0), (1.0, 1e-8), (1e8, 1.0), (1.0, 1e8), (1e-6, 1e6), (1e6, 1e-6), (1e-11, 1e-3)]:
        rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
        rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
        wz, wx = o.backward_mean(az, bz, ax, bx), o.forward_mean(az, bz, ax, bx)
        np.testing.assert_allclose(rz, wz, rtol=0, atol=1e-9 * np.abs(wz).max(), err_msg=f"rz az={az} ax={ax}")
        np.testing.assert_allclose(rx, wx, rtol=0, atol=1e-9 * max(np.abs(wx).max(), 1e-300), err_msg=f"rx az={az} ax={ax}")
        np.testing.assert_allclose(vz, o.backward_variance(az, ax), rtol=1e-6, atol=1e-12 / az)
        np.testing.assert_allclose(vx, o.forward_variance(az, ax), rtol=1e-6, atol=1e-12 / max(ax, 1e-300))


@pytest.mark.parametrize("kind", ["conv", "dft"])
def test_complex_channels_inside_an_ep_model(ctx, kind):
    """ConvChannel(real=False) / DFTChannel(real=False) as nodes of a model: the planar (2,) + shape vectors have to
    reach the scheduler's message buffers (Model._out_shape).  Gaussian prior + linear channel + Gaussian likelihood:
    EP is exact, the posterior mean of the input has a closed form in the Fourier basis."""
    import tramp_amd as T
    rs = np.random.RandomState(5)
    n, nv = 256, 0.05
    z = rs.standard_normal(n) + 1j * rs.standard_normal(n)
    if kind == "conv":
        f = np.zeros(n, dtype=complex)
        f[:4] = rs.standard_normal(4) + 1j * rs.standard_normal(4)
        ch = T.ConvChannel(f, real=False, ctx=ctx)
        w_hat = np.conjugate(np.fft.fft(f))                     # conv_channel.py:42-55: x = ifft(conj(fft(filter)) fft(z))
        x = np.fft.ifft(w_hat * np.fft.fft(z))
    else:
        ch = T.DFTChannel(real=False, ctx=ctx)
        x = np.fft.fft(z, norm="ortho")
    y = x + np.sqrt(nv) * (rs.standard_normal(n) + 1j * rs.standard_normal(n))
    planar = lambda c: np.stack([c.real, c.imag])               # noqa: E731
    model = (T.GaussianPrior(size=(2, n), ctx=ctx) @ T.SISOVariable(id="z") @ ch @ T.SISOVariable(id="x")
             @ T.GaussianLikelihood(y=planar(y).reshape(1, -1), var=nv, ctx=ctx)).to_model()
    assert model.init_shapes() == {"z": (2, n), "x": (2, n)}
    ep = T.ExpectationPropagation(model, ctx=ctx)
    ep.iterate(max_iter=5)
    r = np.asarray(ep.get_variables_data()["z"]["r"]).reshape(2, n)
    if kind == "conv":
        ref = np.fft.ifft(np.conj(w_hat) * np.fft.fft(y) / nv / (1 + np.abs(w_hat)**2 / nv))
    else:
        ref = np.fft.ifft(y, norm="ortho") / (1 + nv)
    np.testing.assert_allclose(r[0] + 1j * r[1], ref, rtol=0, atol=1e-9 * np.abs(ref).max())
