"""GPU parity of the opt-in int8-digit engine of the dense passes (csrc/gemm_sliced.h, tramp_matrix_attach_digits):
the same two products as tests/test_gpu_matvec.py, on v_mfma_i32_16x16x64_i8 with exact int32 accumulation, against
a long-double host product; its error against the f64 matrix-core pass; special values; the LinearChannel switch."""
import ctypes as C

import numpy as np
import pytest

from oracle import tramp_oracle as orc

pytestmark = pytest.mark.gpu


def exact(mat, x, kind):
    m, xx = mat.astype(np.longdouble), x.astype(np.longdouble)
    return xx @ (m.T if kind == "nt" else m)


def run(ctx, T, kind, algo, mat, x, dtype, digits=None, which=3):
    R, Cc = mat.shape
    md, xd = ctx.array(mat.astype(dtype), dtype), ctx.array(x.astype(dtype), dtype)
    if digits is not None:
        ctx.attach_digits(md, digits, which)
    out = ctx.zeros((x.shape[0], R if kind == "nt" else Cc), dtype)
    try:
        ctx._call("tramp_matvec_" + kind, T.device.dtype_code(dtype), algo, R, Cc, x.shape[0], md.ptr, xd.ptr, out.ptr)
        return out.get()
    finally:
        if digits is not None:
            ctx.detach_digits(md)


def data(R, Cc, B, kind, seed=0):
    rs = np.random.RandomState(seed + R * 7 + Cc * 3 + B)
    mat = rs.standard_normal((R, Cc)) / np.sqrt(Cc)
    x = rs.standard_normal((B, Cc if kind == "nt" else R)) * np.where(rs.rand(B, 1) < 0.3, 50.0, 0.2)
    x[:, ::5] *= 1e-3                                          # wide dynamic range inside an instance
    return mat, x


@pytest.mark.parametrize("kind", ["nt", "nn"])
@pytest.mark.parametrize("shape", [(16, 64), (333, 4100), (100, 37), (37, 100), (5, 7), (1000, 500), (48, 20000)])
@pytest.mark.parametrize("B", [1, 2, 3, 50, 64, 70, 130])
@pytest.mark.parametrize("digits", [7, 8])
def test_digit_pass_f64(ctx, kind, shape, B, digits):
    """Worst-case bound of the header: with u = 2^(2 - 8 digits), |error| <= u (4 (digits - 1) k max|row| max|x| +
    max|row| sum|x| + max|x| sum|row|) for the dropped digit products and the rounding of the two operands to
    6 + 8 (digits - 1) bits, plus the f64 sums of the partials."""
    import tramp_amd as T
    mat, x = data(*shape, B, kind)
    before = ctx.sliced_runs()
    out = run(ctx, T, kind, 0, mat, x, np.float64, digits)
    assert ctx.sliced_runs() == before + 1
    ref = exact(mat, x, kind)
    m = mat if kind == "nt" else mat.T                         # logical rows
    k = m.shape[1]
    amax, asum = np.abs(m).max(axis=1)[None, :], np.abs(m).sum(axis=1)[None, :]
    xmax, xsum = np.abs(x).max(axis=1)[:, None], np.abs(x).sum(axis=1)[:, None]
    u = 2.0 ** (2 - 8 * digits)
    bound = u * (4 * (digits - 1) * k * amax * xmax + amax * xsum + xmax * asum) + 2.0 ** -50 * np.abs(ref).astype(np.float64)
    assert np.all(np.abs(out - ref) <= bound + 1e-300), float(np.max(np.abs(out - ref) / bound))
    kdim = shape[1] if kind == "nt" else shape[0]
    if kdim >= 4000:
        # against the f64 matrix-core pass on the same data (instances of very different scale, components spread over
        # three decades - the per-instance scale of the digits is at its worst): the digit error does not grow with k
        # (a fixed 2^(2 - 8 digits) of max|row| sum|x|), a floating-point sum's does.  From a few thousand terms on,
        # 8 digits are more accurate than the f64 pass and 7 digits within a small factor of it.
        f64 = run(ctx, T, kind, 2, mat, x, np.float64)
        assert np.abs(out - ref).max() <= (8.0 if digits == 7 else 1.0) * np.abs(f64 - ref).max()


@pytest.mark.parametrize("kind", ["nt", "nn"])
@pytest.mark.parametrize("shape", [(333, 4100), (100, 37), (1000, 500)])
@pytest.mark.parametrize("B", [9, 64, 65])
def test_digit_pass_f32(ctx, kind, shape, B):
    "f32 operands: 4 digits (30 bits); the result is the f32 rounding of a product accurate far beyond f32"
    import tramp_amd as T
    mat, x = data(*shape, B, kind)
    mat, x = mat.astype(np.float32), x.astype(np.float32)
    out = run(ctx, T, kind, 0, mat, x, np.float32, 0)
    ref = exact(mat, x, kind).astype(np.float64)
    np.testing.assert_allclose(out, ref, rtol=3e-7, atol=3e-7 * np.abs(ref).max())


@pytest.mark.parametrize("kind", ["nt", "nn"])
@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (17, 65), (16, 64), (100, 37), (333, 4100), (48, 20000), (130, 70)])
@pytest.mark.parametrize("B", [1, 3, 64, 70])
@pytest.mark.parametrize("digits", [4, 6, 7, 8])
def test_digit_pass_bit_exact_against_the_integer_restatement(ctx, kind, shape, B, digits):
    """Integer work: the device's digits, its exact int32 sums per split and weight and the order of its f64 operations
    are restated in oracle/tramp_oracle.py:digit_matvec - the two agree bit for bit."""
    import tramp_amd as T
    mat, x = data(*shape, B, kind, seed=digits)
    out = run(ctx, T, kind, 0, mat, x, np.float64, digits)
    ref = orc.digit_matvec(mat if kind == "nt" else mat.T, x, digits)
    assert np.array_equal(out, ref)


def test_forms_and_engine_selection(ctx):
    import tramp_amd as T
    mat, x = data(64, 256, 8, "nt")
    md, xd = ctx.array(mat, np.float64), ctx.array(x, np.float64)
    out = ctx.zeros((8, 64), np.float64)
    dt = T.device.dtype_code(np.float64)
    assert ctx.digits_info(md) == (0, 0, 0)
    with pytest.raises(T._lib.TrampHipError):                   # algo 3 demands a digit copy
        ctx._call("tramp_matvec_nt", dt, 3, 64, 256, 8, md.ptr, xd.ptr, out.ptr)
    ctx.attach_digits(md, 0, 1)                                # the row-dot form only, digits by dtype
    dg, forms, nbytes = ctx.digits_info(md)
    assert (dg, forms) == (7, 1) and nbytes == 4 * 4 * 7 * 1024 + 64 * 8
    n0 = ctx.sliced_runs()
    ctx._call("tramp_matvec_nt", dt, 0, 64, 256, 8, md.ptr, xd.ptr, out.ptr)
    ctx._call("tramp_matvec_nt", dt, 3, 64, 256, 8, md.ptr, xd.ptr, out.ptr)
    assert ctx.sliced_runs() == n0 + 2
    ctx._call("tramp_matvec_nt", dt, 2, 64, 256, 8, md.ptr, xd.ptr, out.ptr)       # forced f64 MFMA
    ctx._call("tramp_matvec_nt", dt, 1, 64, 256, 2, md.ptr, xd.ptr, out.ptr)       # forced GEMV
    y = ctx.array(np.ones((8, 64)), np.float64)
    o2 = ctx.zeros((8, 256), np.float64)
    ctx._call("tramp_matvec_nn", dt, 0, 64, 256, 8, md.ptr, y.ptr, o2.ptr)         # no column form attached: f64 MFMA
    assert ctx.sliced_runs() == n0 + 2
    np.testing.assert_allclose(o2.get(), np.ones((8, 64)) @ mat, rtol=1e-12)
    with pytest.raises(T._lib.TrampHipError):
        ctx.attach_digits(md, 5, 3)
    ctx.attach_digits(md, 8, 3)                                # re-attaching replaces
    assert ctx.digits_info(md)[:2] == (8, 3)
    ctx.detach_digits(md)
    assert ctx.digits_info(md) == (0, 0, 0)
    # releasing the matrix drops its digit copy (the next allocation may get the same address)
    ctx.attach_digits(md, 7, 1)
    ptr = md.ptr
    md.__del__()                                               # what dropping the last reference does
    dg, forms, nbytes = C.c_int(), C.c_int(), C.c_size_t()
    ctx._call("tramp_matrix_digits_info", ptr, C.byref(dg), C.byref(forms), C.byref(nbytes))
    assert dg.value == 0


def test_special_values(ctx):
    "a non-finite entry turns its instance (x) or its output row (matrix) into NaN and nothing else; zeros stay zeros"
    import tramp_amd as T
    mat, x = data(40, 300, 20, "nt")
    x[3, 17] = np.nan
    x[5, 0] = np.inf
    x[7] = 0.0
    mat[11] = 0.0
    out = run(ctx, T, "nt", 0, mat, x, np.float64, 7)
    assert np.all(np.isnan(out[3])) and np.all(np.isnan(out[5]))
    ok = [b for b in range(20) if b not in (3, 5)]
    assert np.all(np.isfinite(out[ok]))
    assert np.all(out[7] == 0.0) and np.all(out[ok][:, 11] == 0.0)
    np.testing.assert_allclose(out[ok], (x[ok] @ mat.T), rtol=1e-12, atol=1e-13 * np.abs(x[ok] @ mat.T).max())
    mat[2, 5] = np.nan
    x2 = np.where(np.isfinite(x), x, 1.0)
    out = run(ctx, T, "nt", 0, mat, x2, np.float64, 7)
    assert np.all(np.isnan(out[:, 2])) and np.all(np.isfinite(np.delete(out, 2, axis=1)))
    # tiny and huge scales: power-of-two scaling is exact
    mat3, x3 = data(32, 128, 5, "nt")
    out = run(ctx, T, "nt", 0, mat3 * 2.0 ** -500, x3 * 2.0 ** 400, np.float64, 8)
    np.testing.assert_allclose(out * 2.0 ** 100, x3 @ mat3.T, rtol=1e-13, atol=1e-14 * np.abs(x3 @ mat3.T).max())


def test_rows_at_the_ends_of_the_exponent_range(ctx):
    """a row whose largest entry is below ~2^-962 made the slicing's multiplier 2^62 / scale overflow (0 * inf =
    NaN digits, wrong FINITE numbers out); the scale is clamped at 2^-960 now, and a maximum of 2^1023 or more makes the
    row non-finite.  Against the integer restatement of the oracle, bit for bit."""
    import tramp_amd as T
    mat, x = data(32, 128, 4, "nt")
    mat = mat.copy()
    mat[3] *= 2.0 ** -1000               # tiny row: scale clamps at 2^-960, digits hold what is left of it
    mat[4] *= 2.0 ** -1060               # subnormal row
    mat[5] = 0.0
    mat[6, 7] = 2.0 ** 1023              # finite, but its scale would be inf
    for S in (7, 8):
        out = run(ctx, T, "nt", 0, mat, x, np.float64, S)
        ref = orc.digit_matvec(mat, x, S)
        assert np.all(np.isnan(out[:, 6])) and np.all(np.isnan(ref[:, 6]))
        keep = [r for r in range(32) if r != 6]
        assert np.all(np.isfinite(out[:, keep]))
        assert np.array_equal(out[:, keep], ref[:, keep])
        assert np.all(out[:, 5] == 0.0)
        exact = x @ mat[keep].T
        assert np.all(np.abs(out[:, keep] - exact) <= 1e-12 * np.abs(exact).max(axis=0, keepdims=True) + 2.0 ** -1000)


def test_digit_copy_does_not_outlive_a_buffer_freed_during_capture(ctx):
    """tramp_buf_free inside a graph capture defers the hipFree; the digit copy keyed by that pointer must be
    out of use at once (a later allocation can get the same address, shape and dtype), and freed with the buffer."""
    import ctypes as C
    import tramp_amd as T
    mat, x = data(48, 192, 3, "nt")
    dt = T.device.dtype_code(np.float64)
    m_d = ctx.array(mat, np.float64)
    ctx.attach_digits(m_d, 7, 1)
    assert ctx.digits_info(m_d)[0] == 7
    ptr = m_d.ptr
    ctx._call("tramp_graph_begin")
    ctx._call("tramp_buf_free", ptr)                 # deferred: the capture is open
    d, w, nb = C.c_int(), C.c_int(), C.c_size_t()
    ctx._call("tramp_matrix_digits_info", ptr, C.byref(d), C.byref(w), C.byref(nb))
    assert d.value == 0, "the digit copy of a buffer released during a capture is still registered"
    g = C.c_void_p()
    ctx._call("tramp_graph_end", C.byref(g))
    if g:
        ctx._call("tramp_graph_destroy", g)
    m_d.ptr = None                                    # the wrapper must not free it a second time
    # a new matrix (quite possibly at the same address): the f64 engine must run, not the old digits
    m2 = ctx.array(-mat, np.float64)
    x_d = ctx.array(x, np.float64)
    out = ctx.empty((x.shape[0], mat.shape[0]), np.float64)
    n0 = ctx.sliced_runs()
    ctx._call("tramp_matvec_nt", dt, 0, mat.shape[0], mat.shape[1], x.shape[0], m2.ptr, x_d.ptr, out.ptr)
    assert ctx.sliced_runs() == n0
    np.testing.assert_allclose(out.get(), -(x @ mat.T), rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("digits", [7, 8])
def test_ep_ensemble_on_the_digit_engine(ctx, digits):
    "the sparse-GLM ensemble of tests/test_gpu_ep.py with LinearChannel(digits=...): same fixed point as the oracle's EP"
    import tramp_amd as T
    N, M, rho, nv, batch = 256, 128, 0.1, 1e-2, 8
    rs = np.random.RandomState(11)
    W = rs.randn(M, N) / np.sqrt(N)
    X = np.stack([orc.gb_sample(rs, N, rho) for _ in range(batch)])
    Y = X @ W.T + np.sqrt(nv) * rs.standard_normal((batch, M))
    lin = T.LinearChannel(W, ctx=ctx, svd="host", digits=digits)
    assert lin.digits == digits
    model = (T.GaussBernoulliPrior(size=N, rho=rho, ctx=ctx) @ T.SISOVariable(id="x") @ lin @ T.SISOVariable(id="z")
             @ T.GaussianLikelihood(y=Y, var=nv, ctx=ctx)).to_model()
    ep = T.ExpectationPropagation(model, ctx=ctx)
    n0 = ctx.sliced_runs()
    ep.iterate(max_iter=8, callback=T.PassCallback(), damping=0.1)
    assert ctx.sliced_runs() >= n0 + 8
    r = ep.get_variables_data()["x"]["r"].reshape(batch, N)
    lf = orc.LinearFactor(W, thin=True)
    for b in range(batch):
        res = orc.ep_chain([orc.GBPriorFactor(N, rho=rho), lf, orc.GaussLikFactor(Y[b], nv)], [N, M], max_iter=8, damping=0.1, tol=0.0)
        np.testing.assert_allclose(r[b], res["r"][0], rtol=1e-6, atol=1e-6 * np.abs(res["r"][0]).max())
    # the steady state captured in a hipGraph (the digit split of x is part of the captured launch sequence) gives the
    # same estimates, bit for bit
    n1 = ctx.sliced_runs()
    ep.iterate(max_iter=8, callback=T.PassCallback(), damping=0.1, graph=True)
    assert ctx.sliced_runs() > n1
    assert np.array_equal(ep.get_variables_data()["x"]["r"].reshape(batch, N), r)
    lin.detach_digits()
    assert lin.digits == 0 and ctx.digits_info(lin._Vt)[0] == 0
