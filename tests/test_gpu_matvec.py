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


# ---- the launch configurations the headline run uses (50 000 x 100 000 x 64: 32 splits; ensembles of more than 64
# instances: sibling chunks interleaved in a one-dimensional grid).  pick_split (csrc/gemm_mfma.hip) wants >= 16 k-steps
# per split and takes the smallest multiple of 8 whose last round of workgroups is >= 90 % full, so these shapes are
# chosen by their block counts: 30 blocks -> 24 splits, 100 blocks -> 32 splits on a 256-CU part.
_BIG = {}


def _big_case(ctx, T, kind, dtype, nblocks, B):
    import ctypes as C
    e = 2 if dtype == np.float64 else 4
    kdim = 8192 if dtype == np.float64 else 16384          # >= 16 k-steps (of 16 / 32 elements) for each of 32 splits
    if kind == "nt":
        R, Cc = 256 * nblocks, kdim
    else:
        R, Cc = kdim, 64 * e * nblocks
    key = (kind, np.dtype(dtype).name, nblocks)
    if key not in _BIG:
        _BIG.clear()                                       # one matrix resident at a time (up to 1.7 GB)
        md = ctx.randn((R, Cc), dtype, seed=11 + nblocks, scale=1.0)
        _BIG[key] = (md, md.get().astype(np.float64))
    md, mat = _BIG[key]
    rs = np.random.RandomState(B)
    x = rs.standard_normal((B, Cc if kind == "nt" else R)).astype(dtype)
    xd = ctx.array(x, dtype)
    out = ctx.zeros((B, R if kind == "nt" else Cc), dtype)
    ctx._call("tramp_matvec_" + kind, T.device.dtype_code(dtype), 2, R, Cc, B, md.ptr, xd.ptr, out.ptr)
    ns, nc, flat = C.c_int(), C.c_int(), C.c_int()
    ctx._call("tramp_matvec_last_launch", C.byref(ns), C.byref(nc), C.byref(flat))
    ref = x.astype(np.float64) @ (mat.T if kind == "nt" else mat)
    return out.get(), ref, ns.value, nc.value, flat.value


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["nt", "nn"])
@pytest.mark.parametrize("nblocks,split64", [(30, 24), (100, 32)])
def test_gemm_mfma_production_launch_shapes(ctx, kind, nblocks, split64, dtype):
    """Both matrix passes at the split-K factors (24, 32) and chunk layouts (one 64-instance chunk, a ragged second
    chunk, two full chunks, three chunks) of the large runs, against float64 NumPy - the small-shape cases above
    never leave 8 splits."""
    import tramp_amd as T
    tol = 1e-12 if dtype == np.float64 else 2e-5
    seen = set()
    for B in (64, 65, 128, 130):
        out, ref, nsplit, nchunk, flat = _big_case(ctx, T, kind, dtype, nblocks, B)
        assert nchunk == (B + 63) // 64
        if B == 64:
            assert nsplit == split64, f"pick_split chose {nsplit} for {nblocks} blocks (expected {split64})"
        assert nsplit % 8 == 0 and nsplit >= 8
        if nchunk > 1:
            assert flat == 1, "sibling chunks were not interleaved (wg_item's one-dimensional launch)"
        seen.add((nsplit, nchunk))
        np.testing.assert_allclose(out, ref, rtol=tol, atol=tol * np.abs(ref).max(), err_msg=f"B={B} split={nsplit}")
    assert len({c for _, c in seen}) == 3
    _BIG.clear()


def test_mfma_deterministic(ctx):
    import tramp_amd as T
    a, _ = run(ctx, T, "nt", 2, 1000, 2048, 64, np.float64)
    b, _ = run(ctx, T, "nt", 2, 1000, 2048, 64, np.float64)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(1, 1), (33, 65), (100, 7), (257, 1030), (2048, 96)])
def test_transpose(ctx, shape, dtype):
    import tramp_amd as T
    R, C = shape
    a = np.random.RandomState(R + C).standard_normal((R, C)).astype(dtype)
    src, dst = ctx.array(a, dtype), ctx.zeros((C, R), dtype)
    ctx._call("tramp_transpose", T.device.dtype_code(dtype), R, C, src, dst)
    assert np.array_equal(dst.get(), a.T)
