"""LinearChannel(W, precompute_svd=False) on the device (csrc/solve.hip: W^T W kept, one Cholesky solve per instance
and call) against the reference's own outputs for that branch (tests/golden/solve.npz,
tramp/channels/linear/linear_channel.py:43-46, 79-82) and against the oracle."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _tol(dtype, c):
    # the wide / rank-deficient cases solve an ill-conditioned system when az is tiny (cond ~ ax s_max / az)
    cond = (c["az"] + c["ax"] * max(c["spectrum"].max(), 0)) / c["az"]
    return (1e-12 if dtype == np.float64 else 2e-6) * max(1.0, cond)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_solve_channel_matches_reference_goldens(ctx, dtype):
    import tramp_amd as T
    cases = load_golden("solve")
    chans = {}
    for c in cases:
        if c["name"] not in chans:
            chans[c["name"]] = T.LinearChannel(c["W"], precompute_svd=False, dtype=dtype, ctx=ctx)
        ch = chans[c["name"]]
        assert isinstance(ch, T.LinearChannel) and type(ch).__name__ == "SolveLinearChannel" and not ch.precompute_svd
        assert ch.rank == c["rank"]
        st = 1e-12 if dtype == np.float64 else 1e-5
        np.testing.assert_allclose(ch.spectrum, c["spectrum"], rtol=0, atol=st * max(1.0, c["spectrum"].max()))
        az, ax, bz, bx = c["az"], c["ax"], c["bz"].astype(dtype), c["bx"].astype(dtype)
        tol = _tol(dtype, c)
        vt = 1e-9 if dtype == np.float64 else 1e-4        # the variances read the device spectrum (f32: W rounded)
        np.testing.assert_allclose(ch.compute_n_eff(az, ax), c["n_eff"], rtol=vt, atol=vt)
        rz, vz = ch.compute_backward_posterior(az, bz, ax, bx)
        rx, vx = ch.compute_forward_posterior(az, bz, ax, bx)
        np.testing.assert_allclose(rz, c["rz"], rtol=0, atol=tol * np.abs(c["rz"]).max())
        np.testing.assert_allclose(rx, c["rx"], rtol=0, atol=tol * max(np.abs(c["rx"]).max(), np.abs(c["rz"]).max()))
        # wide matrices: `singular` is a handful of round-off-sized eigenvalues there (the quirk), so is the variance
        # (eigenvalues of size eps * lambda_max divided by az / ax: both sides compute noise there)
        noise = (1e-13 if dtype == np.float64 else 1e-5) * max(1.0, c["spectrum"].max() / c["az"])
        np.testing.assert_allclose([vz, vx], [c["vz"], c["vx"]], rtol=vt, atol=noise)
        if dtype == np.float32 and c["rank"] < c["W"].shape[1]:
            # Nz > rank: `singular` holds round-off (1e-17 in the reference, 1e-8 once W is rounded to f32), the forward
            # variance built on it is noise of that size and a_new = 1 / v is clipped in one and not in the other
            continue
        a_b, b_b = ch.compute_backward_message(az, bz, ax, bx)
        a_f, b_f = ch.compute_forward_message(az, bz, ax, bx)
        # a_new = 1/v - a cancels when the other direction dominates: compare on the scale of 1/v
        np.testing.assert_allclose(a_b, c["a_bwd"], rtol=vt, atol=vt * (c["az"] + c["a_bwd"]))
        np.testing.assert_allclose(a_f, c["a_fwd"], rtol=vt, atol=vt * (c["ax"] + c["a_fwd"]))
        sb = np.abs(c["rz"]).max() * (c["az"] + c["a_bwd"]) + np.abs(bz).max()
        sf = max(np.abs(c["rx"]).max(), np.abs(c["rz"]).max()) * (c["ax"] + c["a_fwd"]) + np.abs(bx).max()
        np.testing.assert_allclose(b_b, c["b_bwd"], rtol=0, atol=max(tol, vt) * sb)
        np.testing.assert_allclose(b_f, c["b_fwd"], rtol=0, atol=max(tol, vt) * sf)
        lt = 1e-10 if dtype == np.float64 else 1e-4
        np.testing.assert_allclose(ch.compute_log_partition(az, bz, ax, bx), c["logZ"], rtol=max(lt, tol), atol=max(lt, tol))


def test_solve_channel_batched_instances(ctx):
    "An ensemble through one call: every instance with its own (az, ax), against the oracle instance by instance."
    import tramp_amd as T
    rs = np.random.RandomState(5)
    M, N, B = 40, 64, 7
    W = rs.randn(M, N) / np.sqrt(N)
    ch = T.LinearChannel(W, precompute_svd=False, ctx=ctx)
    o = orc.LinearSolveOracle(W)
    az, ax = rs.uniform(0.5, 2, B), rs.uniform(0.5, 2, B)
    bz, bx = rs.randn(B, N), rs.randn(B, M)
    a_f, b_f = ch.compute_forward_message(az, bz, ax, bx)
    a_b, b_b = ch.compute_backward_message(az, bz, ax, bx)
    lz = ch.compute_log_partition(az, bz, ax, bx)
    for b in range(B):
        ra, rb = o.forward_message(az[b], bz[b], ax[b], bx[b])
        np.testing.assert_allclose(a_f[b], ra, rtol=1e-9)
        np.testing.assert_allclose(b_f[b], rb, rtol=0, atol=1e-11 * np.abs(rb).max())
        ra, rb = o.backward_message(az[b], bz[b], ax[b], bx[b])
        np.testing.assert_allclose(a_b[b], ra, rtol=1e-9)
        np.testing.assert_allclose(b_b[b], rb, rtol=0, atol=1e-11 * np.abs(rb).max())
        np.testing.assert_allclose(lz[b], o.log_partition(az[b], bz[b], ax[b], bx[b]), rtol=1e-11)


def test_solve_channel_singular_system_raises(ctx):
    "az = 0 with a rank-deficient W: numpy raises LinAlgError('Singular matrix'); the device reports the failed Cholesky"
    import tramp_amd as T
    rs = np.random.RandomState(2)
    W = rs.randn(6, 3) @ rs.randn(3, 8)
    ch = T.LinearChannel(W, precompute_svd=False, ctx=ctx)
    assert ch.rank == 3
    with pytest.raises(T._lib.TrampHipError, match="singular matrix"):
        ch.compute_backward_posterior(0.0, rs.randn(8), 1.0, rs.randn(6))
    rz, _ = ch.compute_backward_posterior(0.5, rs.randn(8), 1.0, rs.randn(6))      # the handle still works
    assert np.all(np.isfinite(rz))


@pytest.mark.parametrize("damping", [None, 0.3])
def test_ep_with_solve_channel_equals_svd_channel_for_full_column_rank(ctx, damping):
    """Nx >= Nz = rank: the two branches of the reference are the same map, so EP on either channel follows the same
    trajectory (the scheduler runs the solve channel eagerly: no rotated basis, no graph)."""
    import tramp_amd as T
    rs = np.random.RandomState(9)
    N, M, B = 48, 96, 3
    W = rs.randn(M, N) / np.sqrt(N)
    x = rs.randn(B, N) * (rs.rand(B, N) < 0.3)
    y = x @ W.T + 0.1 * rs.randn(B, M)
    res = {}
    for pre in (True, False):
        model = (T.GaussBernoulliPrior(size=N, rho=0.3, ctx=ctx) @ T.SISOVariable(id="x")
                 @ T.LinearChannel(W, precompute_svd=pre, ctx=ctx) @ T.SISOVariable(id="z")
                 @ T.GaussianLikelihood(y=y, var=0.01, ctx=ctx)).to_model()
        ep = T.ExpectationPropagation(model, ctx=ctx)
        ep.iterate(max_iter=40, damping=damping)
        d = ep.get_variables_data()
        res[pre] = (ep.n_iter, np.asarray(d["x"]["r"]), np.asarray(d["x"]["v"]), np.asarray(d["z"]["r"]))
    assert res[True][0] == res[False][0]
    for k in (1, 2, 3):
        np.testing.assert_allclose(res[False][k], res[True][k], rtol=1e-6, atol=1e-8)


def test_solve_channel_groups_of_instances_in_subprocess():
    """More instances than one strided batch of N x N systems holds (scratch limited to 1 MB: 7 instances of a 200 x 200
    system in groups of 3): the loop over groups gives every instance its own solve.  The limit is read once per process."""
    import os
    import subprocess
    import sys
    code = r"""
import sys
sys.path.insert(0, ".")
import numpy as np, tramp_amd as T
from oracle import tramp_oracle as orc
ctx = T.default_context()
rs = np.random.RandomState(6)
M, N, B = 150, 200, 7
W = rs.randn(M, N) / np.sqrt(N)
ch = T.LinearChannel(W, precompute_svd=False, ctx=ctx)
o = orc.LinearSolveOracle(W)
az, ax = rs.uniform(0.5, 2, B), rs.uniform(0.5, 2, B)
bz, bx = rs.randn(B, N), rs.randn(B, M)
a_b, b_b = ch.compute_backward_message(az, bz, ax, bx)
a_f, b_f = ch.compute_forward_message(az, bz, ax, bx)
for b in range(B):
    ra, rb = o.backward_message(az[b], bz[b], ax[b], bx[b])
    assert abs(a_b[b] - ra) < 1e-9 * ra and np.abs(b_b[b] - rb).max() < 1e-10 * np.abs(rb).max(), b
    ra, rb = o.forward_message(az[b], bz[b], ax[b], bx[b])
    assert abs(a_f[b] - ra) < 1e-9 * ra and np.abs(b_f[b] - rb).max() < 1e-10 * np.abs(rb).max(), b
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRAMP_LSOLVE_SCRATCH_MB="1")
    res = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "ok" in res.stdout, res.stdout + res.stderr


def test_adaptive_damping_on_a_tree_model_is_refused(ctx):
    import tramp_amd as T
    from tests.test_gpu_tree import build_tree
    rs = np.random.RandomState(0)
    model = build_tree(T, ctx, dict(shape=(16,), noise_var=0.01, rho=0.1), y=rs.randn(16))
    with pytest.raises(NotImplementedError, match="chain"):
        T.ExpectationPropagation(model, ctx=ctx).iterate(max_iter=2, damping="adaptive")
