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


def test_f64_4x4x4_block_kernels_in_subprocess():
    """The alternative f64 MFMA kernels (v_mfma_f64_4x4x4_4b, TRAMP_MFMA_F64_SHAPE=4; not the
    default, see gemm_mfma.hip) give the same results: run a few shapes in a child process
    because the choice is read once per process."""
    import os
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, ".")
import numpy as np, tramp_amd as T
ctx = T.default_context(); dt = T.device.dtype_code(np.float64)
rs = np.random.RandomState(0)
for (R, C, B) in [(96, 160, 64), (130, 70, 37), (512, 1024, 64), (64, 30, 5), (34, 258, 130)]:
    M = rs.randn(R, C); Xn = rs.randn(B, C); Xt = rs.randn(B, R)
    o1 = ctx.empty((B, R)); o2 = ctx.empty((B, C))
    Md, Xnd, Xtd = ctx.array(M), ctx.array(Xn), ctx.array(Xt)      # keep the buffers alive across the launches
    ctx._call("tramp_matvec_nt", dt, 2, R, C, B, Md.ptr, Xnd.ptr, o1.ptr)
    ctx._call("tramp_matvec_nn", dt, 2, R, C, B, Md.ptr, Xtd.ptr, o2.ptr)
    assert np.abs(o1.get() - Xn @ M.T).max() < 1e-11 * C, (R, C, B, "nt")
    assert np.abs(o2.get() - Xt @ M).max() < 1e-11 * R, (R, C, B, "nn")
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ({}, {"TRAMP_MFMA_RING": "1"}):          # plain 4x4x4 kernels; loader/consumer ring variant (NT)
        env = dict(os.environ, TRAMP_MFMA_F64_SHAPE="4", **extra)
        res = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True,
                             timeout=300)
        assert res.returncode == 0 and "ok" in res.stdout, str(extra) + res.stdout + res.stderr


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(1, 1), (33, 65), (100, 7), (257, 1030), (2048, 96)])
def test_transpose(ctx, shape, dtype):
    import tramp_amd as T
    R, C = shape
    a = np.random.RandomState(R + C).standard_normal((R, C)).astype(dtype)
    src, dst = ctx.array(a, dtype), ctx.zeros((C, R), dtype)
    ctx._call("tramp_transpose", T.device.dtype_code(dtype), R, C, src, dst)
    assert np.array_equal(dst.get(), a.T)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("shape", [(96, 160), (160, 96), (130, 258)])
def test_linear_channel_transposed_copies(ctx, shape, dtype, monkeypatch):
    """Ensembles on the matrix-core path keep Vt^T / Uk^T resident and run the row-dot products as column passes over
    the copies (tramp_linear_set_transposed): same messages as without the copies and as the oracle's LinearChannel
    (tramp/channels/linear/linear_channel.py:69-117)."""
    import tramp_amd as T
    from oracle import tramp_oracle as orc
    M, N = shape
    B = 5
    rs = np.random.RandomState(M)
    W = rs.randn(M, N) / np.sqrt(N)
    az, ax = rs.uniform(0.5, 2, B), rs.uniform(0.5, 2, B)
    bz, bx = rs.standard_normal((B, N)), rs.standard_normal((B, M))
    out = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("TRAMP_LINEAR_TRANSPOSED", flag)
        ch = T.LinearChannel(W, dtype=dtype, ctx=ctx, svd="host")
        fwd = ch.compute_forward_message(az, bz.astype(dtype), ax, bx.astype(dtype))
        bwd = ch.compute_backward_message(az, bz.astype(dtype), ax, bx.astype(dtype))
        assert (ch._Vtt is not None) == (flag == "1") and (ch._Ukt is not None) == (flag == "1")
        if flag == "1":
            assert ch._Vtt.shape == (N, min(M, N)) and ch._Ukt.shape == (min(M, N), M)
            np.testing.assert_array_equal(ch._Vtt.get(), ch._Vt.get().T)
        out[flag] = (fwd, bwd)
    tol = 1e-11 if dtype == np.float64 else 2e-4
    lin = orc.LinearOracle(W)
    for d in (0, 1):
        for b in range(B):
            ref = (lin.forward_message if d == 0 else lin.backward_message)(az[b], bz[b], ax[b], bx[b])
            for flag in ("1", "0"):
                a_new, b_new = out[flag][d]
                np.testing.assert_allclose(a_new[b], ref[0], rtol=max(tol, 1e-9))
                np.testing.assert_allclose(b_new[b], ref[1], rtol=tol, atol=tol * np.abs(ref[1]).max())
