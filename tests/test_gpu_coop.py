"""The one-launch form of the componentwise updates for a FEW large instances (csrc/ew_kernels.hip: ew_coop; BASELINE
configs[1] read literally is ONE instance of 10^6 components): batch x G resident workgroups share the instances, exchange
their block sums through self-validating granules and write the update from their registers.  Against the oracle
(priors/base_prior.py:13-16, likelihoods/base_likelihood.py:25-28, channels/base_channel.py:9-17 of the reference), for
every factor that can take it, f64 and f32, slices that do not divide the instance, damping, posterior output; the
granules reset themselves (same shapes twice, different data; hipGraph-style back-to-back calls), results are run-to-run
identical, and the form actually ran (ctx.coop_runs())."""
import numpy as np
import pytest

from oracle import tramp_oracle as orc
from tests.test_gpu_factors import close, close_a

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tramp_amd
    return tramp_amd


SHAPES = [(1, 1_000_000), (2, 300_002), (3, 70_000), (1, 9000), (5, 41_234)]       # (n above the one-workgroup tile: 4096 f64 / 8192 f32)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("B,n", SHAPES)
def test_priors_and_likelihoods_one_launch(T, ctx, dtype, B, n):
    vn = 2 if dtype == np.float64 else 4
    n -= n % vn
    rs = np.random.RandomState(B + n % 97)
    ax = rs.uniform(0.5, 2.5, B)
    b = (rs.standard_normal((B, n)) * rs.uniform(0.5, 2.0, (B, 1))).astype(dtype)
    y = np.sign(rs.standard_normal((B, n))).astype(dtype)
    cases = [(T.GaussBernoulliPrior(size=n, rho=0.1, dtype=dtype, ctx=ctx), "compute_forward_message",
              lambda i: orc.gb_prior_message(ax[i], b[i].astype(np.float64), rho=0.1)),
             (T.BinaryPrior(size=n, p_pos=0.3, dtype=dtype, ctx=ctx), "compute_forward_message",
              lambda i: orc.ab_new(*orc.binary_prior_posterior(ax[i], b[i].astype(np.float64), p_pos=0.3), ax[i], b[i].astype(np.float64))),
             (T.ReluLikelihood(y=np.maximum(y * np.abs(b), 0), dtype=dtype, ctx=ctx), "compute_backward_message",
              lambda i: orc.relu_lik_message(ax[i], b[i].astype(np.float64), np.maximum(y[i] * np.abs(b[i]), 0).astype(np.float64))),
             (T.SgnLikelihood(y=y, dtype=dtype, ctx=ctx), "compute_backward_message",
              lambda i: orc.sgn_lik_message(ax[i], b[i].astype(np.float64), y[i].astype(np.float64))),
             (T.AbsLikelihood(y=np.abs(b) + 0.1, dtype=dtype, ctx=ctx), "compute_backward_message",
              lambda i: orc.abs_lik_message(ax[i], b[i].astype(np.float64), (np.abs(b[i]) + dtype(0.1)).astype(np.float64)))]
    for f, meth, ref in cases:
        runs0 = ctx.coop_runs()
        a_dev = ctx.empty((B,), np.float64); b_dev = ctx.empty((B, n), dtype)
        getattr(f, meth)(ctx.scalars(ax), ctx.array(b, dtype), out_a=a_dev, out_b=b_dev)
        assert ctx.coop_runs() == runs0 + 1, f"{type(f).__name__}: the one-launch form did not run"
        a1, b1 = a_dev.get().copy(), b_dev.get().copy()
        getattr(f, meth)(ctx.scalars(ax), ctx.array(b, dtype), out_a=a_dev, out_b=b_dev)      # granules reset themselves
        assert np.array_equal(a1, a_dev.get()) and np.array_equal(b1, b_dev.get())                 # and the fold is reproducible
        if ref is None:
            continue
        for i in range(B):
            a_ref, b_ref = ref(i)
            close_a(a1[i], a_ref, ax[i], dtype)
            close(b1[i], b_ref, dtype, scale=10 * max(np.abs(b_ref).max(), 1.0))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["sgn", "abs", "relu"])
def test_pwl_channels_one_launch(T, ctx, kind, dtype):
    B, n = 2, 250_000
    rs = np.random.RandomState(3)
    az, ax = rs.uniform(0.8, 2.0, B), rs.uniform(0.5, 1.5, B)
    bz = (rs.standard_normal((B, n)) * 1.5).astype(dtype); bx = rs.standard_normal((B, n)).astype(dtype)
    bz[1, 7] = 40.0                                   # past the erfcx table: the wave takes the region-by-region form in place
    ch = dict(sgn=T.SgnChannel, abs=T.AbsChannel, relu=T.ReluChannel)[kind](dtype=dtype, ctx=ctx)
    for direction, fn, a_in in (("fwd", ch.compute_forward_message, ax), ("bwd", ch.compute_backward_message, az)):
        runs0 = ctx.coop_runs()
        a_dev = ctx.empty((B,), np.float64); b_dev = ctx.empty((B, n), dtype)
        fn(ctx.scalars(az), ctx.array(bz, dtype), ctx.scalars(ax), ctx.array(bx, dtype), out_a=a_dev, out_b=b_dev)
        assert ctx.coop_runs() == runs0 + 1
        a_new, b_new = a_dev.get(), b_dev.get()
        for i in range(B):
            with np.errstate(all="ignore"):
                a_ref, b_ref = orc.pwl_channel_message(kind, direction, az[i], bz[i].astype(np.float64), ax[i], bx[i].astype(np.float64))
            close_a(a_new[i], a_ref, a_in[i], dtype)
            close(b_new[i], b_ref, dtype, scale=10 * max(np.abs(b_ref).max(), np.abs(bz[i]).max(), 1.0))


def test_one_launch_damping_posterior_and_nan(T, ctx):
    B, n, d = 2, 123_456, 0.3
    rs = np.random.RandomState(9)
    ax = rs.uniform(0.5, 2.0, B); bx = rs.standard_normal((B, n)) * 2
    old_a = rs.uniform(0.5, 2.0, B); old_b = rs.standard_normal((B, n))
    p = T.GaussBernoulliPrior(size=n, rho=0.2, ctx=ctx)
    oa = ctx.scalars(old_a); ob = ctx.array(old_b)
    runs0 = ctx.coop_runs()
    p.compute_forward_message(ctx.scalars(ax), ctx.array(bx), out_a=oa, out_b=ob, damping=d, old_a=oa, old_b=ob)   # in place
    assert ctx.coop_runs() == runs0 + 1
    for i in range(B):
        a_ref, b_ref = orc.gb_prior_message(ax[i], bx[i], rho=0.2)
        close_a(oa.get()[i], d * old_a[i] + (1 - d) * a_ref, ax[i], np.float64)
        close(ob.get()[i], d * old_b[i] + (1 - d) * b_ref, np.float64, scale=10 * np.abs(b_ref).max())
    r, v = p.compute_forward_posterior(ax, bx)                    # posterior mean wanted: the r output of the same kernel
    for i in range(B):
        r_ref, v_ref = orc.gb_prior_posterior(ax[i], bx[i], rho=0.2)
        close(np.asarray(r)[i], r_ref, np.float64); close(np.asarray(v)[i], v_ref, np.float64)
    bx[1, 5] = np.nan
    a_new, _ = p.compute_forward_message(ax, bx)
    assert np.isfinite(a_new[0]) and np.isnan(a_new[1])            # a NaN stays in its own instance


def test_one_launch_form_replays_inside_a_graph(T, ctx):
    """The bank parity lives in a device word that the launch itself advances: a captured iteration replayed any number
    of times (odd count of one-launch updates per iteration) gives the eager loop's result bit for bit."""
    N, B = 30_000, 3
    rs = np.random.RandomState(5)
    y = np.sign(rs.standard_normal((B, N)))
    out = []
    for use_graph in (False, True):
        model = (T.GaussBernoulliPrior(size=N, rho=0.3, ctx=ctx) @ T.SISOVariable(id="x")
                 @ T.GaussianChannel(var=0.5, ctx=ctx) @ T.SISOVariable(id="z")
                 @ T.SgnLikelihood(y=y, ctx=ctx)).to_model()
        ep = T.ExpectationPropagation(model, ctx=ctx)
        runs0 = ctx.coop_runs()
        ep.iterate(max_iter=9, damping=0.2, callback=T.EarlyStoppingEP(tol=0.0), graph=use_graph)
        assert ctx.coop_runs() > runs0
        d = ep.get_variables_data()
        out.append((d["x"]["r"].copy(), np.copy(d["x"]["v"]), d["z"]["r"].copy()))
    for a, b in zip(*out):
        assert np.array_equal(a, b)
    assert np.all(np.isfinite(out[0][0])) and not ctx.resident_gave_up()


def test_one_launch_form_gives_up_cleanly():
    """A block sum that never arrives (forced: workgroup 2 of instance 0 does not publish, deadline 100 ms): the kernel
    ends, instance 0 is loudly NaN, the others are right, the host-visible error word is up and the context takes the
    tile kernels from then on (the same contract as the resident pipeline's, csrc/ew_resident.h)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRAMP_EW_RESIDENT_TIMEOUT_MS="100", TRAMP_EW_COOP_DBG_SKIP="3")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_coop_giveup_worker.py")], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["coop_ran"] and out["gave_up"] and out["second_call_on_tiles"], out
    assert out["nan_instances_first"] == [1, 0, 0] and out["nan_instances_second"] == 0, out
    assert out["finite_agree"] and out["oracle"], out
