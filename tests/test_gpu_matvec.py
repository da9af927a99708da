"""GPU parity of the two matrix primitives behind LinearChannel (K8): VALU GEMV and MFMA paths
against NumPy, over ragged shapes and ensemble sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(ctx, T, kind, algo, R, C, B, dtype):
    rs = np.random.RandomState(R * 7 + C * 3 + B)
    mat = rs.standard_normal((R, C)).astype(dtype)
    x = rs.standard_normal((B, C if kind == "nt" else R)).astype(dtype)
    md, xd = ctx.array(mat, dtype), ctx.array(x, dtype)
    out = ctx.zeros((B, R if kind == "nt" else C), dtype)
    ctx._call("tramp_matvec_" + kind, T.device.dtype_code(dtype), algo, R, C, B, md.ptr, xd.ptr, out.ptr)
    ref = x.astype(np.float64) @ (mat.T if kind == "nt" else mat).astype(np.float64)
    return out.get(), ref


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["nt", "nn"])
@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (64, 96), (130, 257), (257, 130), (1000, 512), (40, 4096)])
@pytest.mark.parametrize("B", [1, 2, 3, 8, 9])
def test_gemv_valu(ctx, kind, shape, B, dtype):
    import tramp_amd as T
    out, ref = run(ctx, T, kind, 1, shape[0], shape[1], B, dtype)
    tol = 1e-12 if dtype == np.float64 else 2e-5
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol * np.abs(ref).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["nt", "nn"])
@pytest.mark.parametrize("shape", [(16, 16), (32, 32), (128, 256), (130, 288), (288, 130), (1000, 512), (64, 4096),
                                   (4096, 64), (50, 1600), (500, 1000), (1000, 500), (37, 102), (102, 38), (5, 7)])
@pytest.mark.parametrize("B", [1, 9, 16, 17, 33, 64, 65, 130])
def test_gemm_mfma(ctx, kind, shape, B, dtype):
    import tramp_amd as T
    R, C = shape
    e = 2 if dtype == np.float64 else 4
    if kind == "nt" and C % e or kind == "nn" and (R % e or C % e):
        with pytest.raises(T._lib.TrampHipError):
            run(ctx, T, kind, 2, R, C, B, dtype)      # algo=2 forces MFMA: unsupported shape
        return
    out, ref = run(ctx, T, kind, 2, R, C, B, dtype)
    tol = 1e-12 if dtype == np.float64 else 2e-5
    np.testing.assert_allclose(out, ref, rtol=tol, atol=tol * np.abs(ref).max())


def test_mfma_deterministic(ctx):
    import tramp_amd as T
    a, _ = run(ctx, T, "nt", 2, 1000, 2048, 64, np.float64)
    b, _ = run(ctx, T, "nt", 2, 1000, 2048, 64, np.float64)
    assert np.array_equal(a, b)
