"""GPU parity of the resident single-pass factor updates (tramp_amd/csrc/ew_resident.h) through the C ABI.

Large batches of large instances take a pipeline of resident workgroups that hands tile sums to combiner workgroups
through self-validating 8-byte granules.  What can go wrong there that the tile kernels cannot get wrong: a granule of
a PREVIOUS launch read as fresh (the same shapes twice with different data), the overlapping last tile of an instance
(ragged tile counts; outputs aliasing inputs), per-instance scalars crossing instances (different precisions, one NaN
instance), the support count's second slot array, damping against the buffer being overwritten.  Every case also
asserts that the resident form actually ran (ctx.resident_runs()).

Reference: tramp/base.py:245-255 (compute_ab_new), priors/gauss_bernoulli_prior.py:70-77,
priors/map_L1_norm_prior.py:66-71, restated in oracle/tramp_oracle.py.
fp64: 1e-9 relative, fp32: 1e-3; counts exact.
"""
import numpy as np
import pytest

from oracle import tramp_oracle as orc
from tests.test_gpu_factors import close, close_a

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tramp_amd
    return tramp_amd


def num_cu(ctx):
    "compute units of the device: parsed from tramp_device_name's '... (gfx950, 256 CUs)'"
    return int(ctx.device_name().rsplit(",", 1)[1].split()[0])


def shape_for(ctx, dtype, ragged=True):
    """(batch, n): enough 16 KiB tiles for the pipeline's minimum (4 rounds of 2 x CUs - 8 tiles), instances that
    are not a whole number of tiles (the last tile overlaps its predecessor) unless ragged=False."""
    comps = 2048 if dtype == np.float64 else 4096          # components per tile
    tiles = 7
    n = tiles * comps - (comps // 2 + 2 * (16 // np.dtype(dtype).itemsize) if ragged else 0)
    need = 4 * (2 * num_cu(ctx) - 8) + 16
    return (need + tiles - 1) // tiles, n


def gb_reference(ax, bx, rho):
    out_a, out_b = [], []
    for a, b in zip(ax, bx):
        an, bn = orc.gb_prior_message(a, b.astype(np.float64), rho=rho)
        out_a.append(an); out_b.append(bn)
    return np.array(out_a), np.stack(out_b)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ragged", [True, False])
def test_gb_prior_resident_vs_oracle_and_no_stale_granules(T, ctx, dtype, ragged):
    B, n = shape_for(ctx, dtype, ragged)
    rs = np.random.RandomState(11)
    p = T.GaussBernoulliPrior(size=n, rho=0.07, dtype=dtype, ctx=ctx)
    a_dev = ctx.empty((B,), np.float64); b_dev = ctx.empty((B, n), dtype)
    prev = None
    for trial in range(3):          # same shapes, same workspace, DIFFERENT data: a granule left by the previous launch would show
        ax = rs.uniform(0.3, 3.0, B)
        bx = (rs.standard_normal((B, n)) * rs.uniform(0.5, 3.0, (B, 1))).astype(dtype)
        runs0 = ctx.resident_runs()
        p.compute_forward_message(ctx.scalars(ax), ctx.array(bx, dtype), out_a=a_dev, out_b=b_dev)
        assert ctx.resident_runs() == runs0 + 1, "the resident single-pass form did not run"
        a_ref, b_ref = gb_reference(ax, bx, 0.07)
        a_new, b_new = a_dev.get(), b_dev.get()
        for i in range(B):
            close_a(a_new[i], a_ref[i], ax[i], dtype)
            close(b_new[i], b_ref[i], dtype, scale=10 * np.abs(b_ref[i]).max())
        if prev is not None:
            assert not np.array_equal(prev, a_new)
        prev = a_new.copy()
    # run-to-run reproducible: fixed-order fold of the tile sums, no float atomics
    a1 = a_dev.get().copy(); b1 = b_dev.get().copy()
    p.compute_forward_message(ctx.scalars(ax), ctx.array(bx, dtype), out_a=a_dev, out_b=b_dev)
    assert np.array_equal(a1, a_dev.get()) and np.array_equal(b1, b_dev.get())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_l1_prior_resident_count_and_message(T, ctx, dtype):
    B, n = shape_for(ctx, dtype)
    rs = np.random.RandomState(5)
    gamma = 0.8
    ax = rs.uniform(0.5, 2.0, B)
    bx = (rs.standard_normal((B, n)) * 1.5).astype(dtype)
    l1 = T.MAP_L1NormPrior(size=n, gamma=gamma, dtype=dtype, ctx=ctx)
    a_dev = ctx.empty((B,), np.float64); b_dev = ctx.empty((B, n), dtype); cnt = ctx.empty((B,), np.uint64)
    bx_d = ctx.array(bx, dtype); ax_d = ctx.scalars(ax)
    runs0 = ctx.resident_runs()
    ctx._call("tramp_l1_prior_fwd", l1._dt, n, B, ax_d.ptr, bx_d.ptr, float(gamma), 0.0, None, None,
              a_dev.ptr, b_dev.ptr, None, None, cnt.ptr, None)
    assert ctx.resident_runs() == runs0 + 1
    a_new, b_new, c = a_dev.get(), b_dev.get(), cnt.get()
    for i in range(B):
        # the support test on the operands of the working precision (map_L1_norm_prior.py:16-17)
        assert int(c[i]) == int((np.abs(bx[i]) > dtype(gamma)).sum())
        a_ref, b_ref = orc.l1_prior_message(ax[i], bx[i].astype(np.float64), gamma=gamma)
        close_a(a_new[i], a_ref, ax[i], dtype)
        close(b_new[i], b_ref, dtype, scale=10 * np.abs(b_ref).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_resident_damping_in_place_and_aliased_input(T, ctx, dtype):
    """damping against the buffer that is being overwritten (out_b = old_b: what the EP scheduler does), then the
    update written over its own input (out_b = bx): every piece is read and written by exactly one tile"""
    B, n = shape_for(ctx, dtype)
    rs = np.random.RandomState(9)
    d = 0.3
    ax = rs.uniform(0.5, 2.0, B); bx = (rs.standard_normal((B, n)) * 2).astype(dtype)
    old_a = rs.uniform(0.5, 2.0, B); old_b = rs.standard_normal((B, n)).astype(dtype)
    p = T.GaussBernoulliPrior(size=n, rho=0.2, dtype=dtype, ctx=ctx)
    oa = ctx.scalars(old_a); ob = ctx.array(old_b, dtype)
    runs0 = ctx.resident_runs()
    p.compute_forward_message(ctx.scalars(ax), ctx.array(bx, dtype), out_a=oa, out_b=ob, damping=d, old_a=oa, old_b=ob)
    assert ctx.resident_runs() == runs0 + 1
    a_ref, b_ref = gb_reference(ax, bx, 0.2)
    a_new, b_new = oa.get(), ob.get()
    for i in range(B):
        close_a(a_new[i], d * old_a[i] + (1 - d) * a_ref[i], ax[i], dtype)
        close(b_new[i], d * old_b[i].astype(np.float64) + (1 - d) * b_ref[i], dtype, scale=10 * np.abs(b_ref[i]).max())
    bio = ctx.array(bx, dtype); ao = ctx.empty((B,), np.float64)
    p.compute_forward_message(ctx.scalars(ax), bio, out_a=ao, out_b=bio)
    assert ctx.resident_runs() == runs0 + 2
    b_new = bio.get()
    for i in range(B):
        close(b_new[i], b_ref[i], dtype, scale=10 * np.abs(b_ref[i]).max())


def test_resident_nan_stays_in_its_instance(T, ctx):
    dtype = np.float64
    B, n = shape_for(ctx, dtype)
    rs = np.random.RandomState(2)
    bx = rs.standard_normal((B, n))
    bad = [1, B - 2]
    bx[bad[0], n - 3] = np.nan              # inside the overlapped last tile
    bx[bad[1], 17] = np.inf
    p = T.GaussBernoulliPrior(size=n, rho=0.2, ctx=ctx)
    runs0 = ctx.resident_runs()
    a_new, _ = p.compute_forward_message(np.full(B, 1.1), bx)
    assert ctx.resident_runs() == runs0 + 1
    assert np.isnan(a_new[bad]).all() and np.isfinite(np.delete(a_new, bad)).all()


def test_shapes_the_window_does_not_cover_take_the_tile_kernels(T, ctx):
    "one instance larger than the pipeline's window; a batch too small to fill it; odd n: same results, other kernels"
    rs = np.random.RandomState(4)
    p_args = dict(rho=0.1, ctx=ctx)
    for B, n in [(1, 2048 * (2 * num_cu(ctx) + 40)), (2, 70001), (3, 20000)]:
        bx = rs.standard_normal((B, n))
        runs0 = ctx.resident_runs()
        a_new, b_new = T.GaussBernoulliPrior(size=n, **p_args).compute_forward_message(np.full(B, 1.2), bx)
        assert ctx.resident_runs() == runs0
        for i in range(B):
            a_ref, b_ref = orc.gb_prior_message(1.2, bx[i], rho=0.1)
            close(a_new[i], a_ref); close(b_new[i], b_ref, scale=10 * np.abs(b_ref).max())


def sgn_reference(az, bz, y):
    out_a, out_b = [], []
    for a, b, yy in zip(az, bz, y):
        an, bn = orc.sgn_lik_message(a, b.astype(np.float64), yy.astype(np.float64))
        out_a.append(an); out_b.append(bn)
    return np.array(out_a), np.stack(out_b)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sign_observation_packs_one_bit_per_component(T, ctx, dtype):
    rs = np.random.RandomState(3)
    for B, n in [(3, 1), (2, 63), (2, 64), (3, 65), (2, 1000), (1, 4099)]:
        y = np.where(rs.standard_normal((B, n)) > 0, 1.0, -1.0).astype(dtype)
        lk = T.SgnLikelihood(y=y, dtype=dtype, ctx=ctx)
        assert lk._bits is not None and lk._bits.shape == (B, ((n + 63) // 64) * 8)
        bits = np.unpackbits(lk._bits.get(), axis=1, bitorder="little")
        assert np.array_equal(bits[:, :n], (y > 0).astype(np.uint8)) and not bits[:, n:].any()
    # an observation outside {-1, +1} (np.sign(0) = 0): no packed form, the update reads y itself
    y0 = np.array([[1.0, -1.0, 0.0, 1.0]], dtype=dtype)
    assert T.SgnLikelihood(y=y0, dtype=dtype, ctx=ctx)._bits is None


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sgn_likelihood_resident_on_packed_observation(T, ctx, dtype):
    B, n = shape_for(ctx, dtype)
    rs = np.random.RandomState(13)
    y = np.where(rs.standard_normal((B, n)) > 0, 1.0, -1.0).astype(dtype)
    lk = T.SgnLikelihood(y=y, dtype=dtype, ctx=ctx)
    a_dev = ctx.empty((B,), np.float64); b_dev = ctx.empty((B, n), dtype)
    for trial in range(2):
        az = rs.uniform(0.4, 2.5, B)
        bz = (rs.standard_normal((B, n)) * 1.5).astype(dtype)
        runs0 = ctx.resident_runs()
        lk.compute_backward_message(ctx.scalars(az), ctx.array(bz, dtype), out_a=a_dev, out_b=b_dev)
        assert ctx.resident_runs() == runs0 + 1
        a_ref, b_ref = sgn_reference(az, bz, y)
        a_new, b_new = a_dev.get(), b_dev.get()
        for i in range(B):
            close_a(a_new[i], a_ref[i], az[i], dtype)
            close(b_new[i], b_ref[i], dtype, scale=10 * np.abs(b_ref[i]).max())
    # the same update on the unpacked observation (y handed in again: the tile kernels or the resident 3T form)
    a2, b2 = lk._message(ctx.scalars(az), ctx.array(bz, dtype), y=ctx.array(y, dtype))[2:4]
    close(a2.get(), a_new, dtype); close(b2.get(), b_new, dtype, scale=10 * np.abs(b_new).max())
    # damped, in place
    d = 0.4
    old_a = rs.uniform(0.5, 2.0, B); old_b = rs.standard_normal((B, n)).astype(dtype)
    oa = ctx.scalars(old_a); ob = ctx.array(old_b, dtype)
    lk.compute_backward_message(ctx.scalars(az), ctx.array(bz, dtype), out_a=oa, out_b=ob, damping=d, old_a=oa, old_b=ob)
    got = ob.get()
    for i in range(B):
        close(got[i], d * old_b[i].astype(np.float64) + (1 - d) * b_ref[i], dtype, scale=10 * np.abs(b_ref[i]).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_abs_likelihood_resident(T, ctx, dtype):
    B, n = shape_for(ctx, dtype)
    rs = np.random.RandomState(17)
    y = np.abs(rs.standard_normal((B, n))).astype(dtype)
    az = rs.uniform(0.4, 2.5, B)
    bz = (rs.standard_normal((B, n)) * 1.5).astype(dtype)
    lk = T.AbsLikelihood(y=y, dtype=dtype, ctx=ctx)
    runs0 = ctx.resident_runs()
    a_new, b_new = lk.compute_backward_message(az, bz)
    assert ctx.resident_runs() == runs0 + 1
    for i in range(B):
        a_ref, b_ref = orc.abs_lik_message(az[i], bz[i].astype(np.float64), y[i].astype(np.float64))
        close_a(a_new[i], a_ref, az[i], dtype)
        close(b_new[i], b_ref, dtype, scale=10 * np.abs(b_ref).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_binary_prior_and_relu_likelihood_resident(T, ctx, dtype):
    """The two further separable updates on the resident pipeline (BinaryPrior: priors/binary_prior.py:27,52-58;
    ReluLikelihood: likelihoods/piecewise_linear_likelihood.py:9-63,279-291) against the oracle, the form asserted."""
    B, n = shape_for(ctx, dtype)
    rs = np.random.RandomState(23)
    ax = rs.uniform(0.4, 2.5, B)
    bx = (rs.standard_normal((B, n)) * 1.2).astype(dtype)
    pr = T.BinaryPrior(size=n, p_pos=0.3, dtype=dtype, ctx=ctx)
    runs0 = ctx.resident_runs()
    a_new, b_new = pr.compute_forward_message(ax, bx)
    assert ctx.resident_runs() == runs0 + 1
    for i in range(0, B, 7):
        r, v = orc.binary_prior_posterior(ax[i], bx[i].astype(np.float64), p_pos=0.3)
        a_ref, b_ref = orc.ab_new(r, v, ax[i], bx[i].astype(np.float64))
        close_a(a_new[i], a_ref, ax[i], dtype)
        close(b_new[i], b_ref, dtype, scale=10 * np.abs(b_ref).max())
    z = rs.standard_normal((B, n))
    y = np.maximum(z, 0.0).astype(dtype)
    az = rs.uniform(0.4, 2.5, B)
    bz = (rs.standard_normal((B, n)) * 1.5).astype(dtype)
    lk = T.ReluLikelihood(y=y, dtype=dtype, ctx=ctx)
    runs0 = ctx.resident_runs()
    a_new, b_new = lk.compute_backward_message(az, bz)
    assert ctx.resident_runs() == runs0 + 1
    for i in range(0, B, 7):
        a_ref, b_ref = orc.relu_lik_message(az[i], bz[i].astype(np.float64), y[i].astype(np.float64))
        close_a(a_new[i], a_ref, az[i], dtype)
        close(b_new[i], b_ref, dtype, scale=10 * np.abs(b_ref).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_relu_likelihood_lean_form_redoes_what_it_cannot_finish(T, ctx, dtype):
    """ReluLikelihood on the pipeline is the lean form (csrc/ew_ops.h ReluLikOp<T, true>: the flat region's half line from
    the one-value erfcx table, no control flow).  A flat component whose message sits far inside z > 0 (erfcx argument
    past the table) sends its instance to the tile kernels - same result as the oracle; an observation no region holds
    (y < 0: piecewise_linear_likelihood.py:36-63, softmax of (-inf, -inf)) or a NaN message makes its own instance NaN
    and leaves the others alone."""
    B, n = shape_for(ctx, dtype)
    rs = np.random.RandomState(29)
    y = np.maximum(rs.standard_normal((B, n)), 0.0).astype(dtype)
    az = rs.uniform(0.4, 2.5, B)
    bz = (rs.standard_normal((B, n)) * 1.5).astype(dtype)
    flat = np.flatnonzero(y[3] == 0)[:3]
    bz[3, flat] = [60.0, 25.0, -80.0]                  # two beyond the table's upper end, one far below its lower end
    y[5, 7] = -1.0
    bz[9, np.flatnonzero(y[9] == 0)[0]] = np.nan
    lk = T.ReluLikelihood(y=y, dtype=dtype, ctx=ctx)
    runs0 = ctx.resident_runs()
    a_new, b_new = lk.compute_backward_message(az, bz)
    assert ctx.resident_runs() == runs0 + 1
    assert np.isnan(a_new[5]) and np.isnan(a_new[9]) and np.all(np.isfinite(np.delete(a_new, [5, 9])))
    for i in (0, 3, 4, 6, B - 1):
        a_ref, b_ref = orc.relu_lik_message(az[i], bz[i].astype(np.float64), y[i].astype(np.float64))
        close_a(a_new[i], a_ref, az[i], dtype)
        close(b_new[i], b_ref, dtype, scale=10 * np.abs(b_ref).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_resident_pipeline_on_seeded_random_shapes_around_its_limits(T, ctx, dtype):
    """Shapes drawn around the eligibility limits of the pipeline (tiles per instance from 1 to past the window, batches
    that just fill or just miss the minimum number of rounds, instance lengths that end inside a tile): whatever form
    runs (the context says which), every instance agrees with the oracle and nothing is left from a previous shape."""
    rs = np.random.RandomState(20260)
    comps = 2048 if dtype == np.float64 else 4096          # components per 16 KiB tile
    vn = 2 if dtype == np.float64 else 4
    ps = 2 * num_cu(ctx) - 8
    took = {True: 0, False: 0}
    for trial in range(14):
        tiles = int(rs.choice([1, 2, 3, 5, 9, 17, 33, ps - 1, ps + 1, ps + 2]))
        n = tiles * comps - int(rs.randint(0, comps // vn)) * vn          # ends inside the last tile, whole 16-byte pieces
        n = max(n, vn)
        want = 4 * ps + int(rs.randint(-2, 3)) * tiles                    # rounds just under / over the minimum
        B = int(np.clip(want // tiles + int(rs.randint(0, 2)), 1, max(1, (96 << 20) // (n * np.dtype(dtype).itemsize))))
        ax = rs.uniform(0.3, 3.0, B)
        bx = (rs.standard_normal((B, n)) * rs.uniform(0.5, 3.0, (B, 1))).astype(dtype)
        covers = ctx.resident_covers(dtype, n, B)
        runs0 = ctx.resident_runs()
        a_new, b_new = T.GaussBernoulliPrior(size=n, rho=0.07, dtype=dtype, ctx=ctx).compute_forward_message(ax, bx)
        assert (ctx.resident_runs() == runs0 + 1) == covers, (tiles, n, B, covers)
        took[covers] += 1
        for i in sorted(set([0, B // 2, B - 1])):
            a_ref, b_ref = orc.gb_prior_message(ax[i], bx[i].astype(np.float64), rho=0.07)
            close_a(a_new[i], a_ref, ax[i], dtype)
            close(b_new[i], b_ref, dtype, scale=10 * np.abs(b_ref).max())
    assert took[True] >= 2 and took[False] >= 2, took


# ---- the two-region channels K7 on the pipeline (round 5: moment form + one-value erfcx table, csrc/ew_resident_pwl.hip)
PWL_CLS = dict(sgn="SgnChannel", abs="AbsChannel", relu="ReluChannel")


def pwl_reference(kind, direction, az, bz, ax, bx):
    out_a, out_b = [], []
    for i in range(len(az)):
        with np.errstate(all="ignore"):
            an, bn = orc.pwl_channel_message(kind, direction, az[i], bz[i].astype(np.float64), ax[i], bx[i].astype(np.float64))
        out_a.append(an); out_b.append(bn)
    return np.array(out_a), np.stack(out_b)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("kind", ["sgn", "abs", "relu"])
def test_pwl_channels_resident_vs_oracle(T, ctx, kind, dtype):
    """Both directions of Sgn / Abs / Relu on the resident pipeline (piecewise_linear_channel.py:27-53 through
    channels/base_channel.py:9-17), per-instance precisions, the same shapes twice with different data; the resident form
    must actually run, and agree with the tile kernels (TRAMP_EW_RESIDENT_K7=0 is what every other K7 test covers)."""
    B, n = shape_for(ctx, dtype)
    rs = np.random.RandomState(21)
    ch = getattr(T, PWL_CLS[kind])(dtype=dtype, ctx=ctx)
    a_dev = ctx.empty((B,), np.float64); b_dev = ctx.empty((B, n), dtype)
    for trial in range(2):
        az, ax = rs.uniform(0.5, 3.0, B), rs.uniform(0.3, 2.0, B)
        bz = (rs.standard_normal((B, n)) * rs.uniform(0.5, 2.0, (B, 1))).astype(dtype)
        bx = (rs.standard_normal((B, n)) * rs.uniform(0.5, 2.0, (B, 1))).astype(dtype)
        for direction, fn, a_in, tgt in (("fwd", ch.compute_forward_message, ax, bx), ("bwd", ch.compute_backward_message, az, bz)):
            runs0 = ctx.resident_runs()
            fn(ctx.scalars(az), ctx.array(bz, dtype), ctx.scalars(ax), ctx.array(bx, dtype), out_a=a_dev, out_b=b_dev)
            assert ctx.resident_runs() == runs0 + 1, "the resident single-pass form did not run"
            a_ref, b_ref = pwl_reference(kind, direction, az, bz, ax, bx)
            a_new, b_new = a_dev.get(), b_dev.get()
            for i in range(B):
                close_a(a_new[i], a_ref[i], a_in[i], dtype)
                # b_new = r (a_in + a_new) - b: conditioned on |r| (a_in + a_new) and |b|
                sc = 10 * max(np.abs(b_ref[i]).max(), np.abs(tgt[i]).max().astype(np.float64))
                close(b_new[i], b_ref[i], dtype, scale=sc)


@pytest.mark.parametrize("kind", ["sgn", "abs", "relu"])
def test_pwl_resident_flagged_instances_are_redone(T, ctx, kind):
    """A component outside the moment form's domain (an erfcx argument below -5 or above 8, a message within 2 % of the
    reference's log_Phi mask threshold, an infinity, a NaN) flags its instance; the redo kernel runs that instance again
    on the region-by-region form.  The flagged instances must equal what the tile kernels give (same evaluation), the
    others the oracle; a NaN stays in its own instance."""
    dtype = np.float64
    B, n = shape_for(ctx, dtype)
    rs = np.random.RandomState(3)
    az, ax = rs.uniform(0.8, 2.0, B), rs.uniform(0.5, 1.5, B)
    bz = rs.standard_normal((B, n)); bx = rs.standard_normal((B, n))
    flagged = {2: (5, 14.0), 7: (n - 3, -11.0), B - 3: (100, 29.8 * np.sqrt(az[B - 3])), B - 1: (9, np.nan), 11: (n // 2, np.inf)}
    for i, (j, val) in flagged.items():
        bz[i, j] = val
    ch = getattr(T, PWL_CLS[kind])(ctx=ctx)
    for direction, fn in (("fwd", ch.compute_forward_message), ("bwd", ch.compute_backward_message)):
        a_dev = ctx.empty((B,), np.float64); b_dev = ctx.empty((B, n), dtype)
        runs0 = ctx.resident_runs()
        fn(ctx.scalars(az), ctx.array(bz, dtype), ctx.scalars(ax), ctx.array(bx, dtype), out_a=a_dev, out_b=b_dev)
        assert ctx.resident_runs() == runs0 + 1
        a_res, b_res = a_dev.get().copy(), b_dev.get().copy()
        # the tile kernels on the same data: instance by instance below the pipeline's window (a batch of one)
        for i in sorted(flagged):
            a1, b1 = fn(az[i], bz[i], ax[i], bx[i])
            if np.isnan(a1):
                assert np.isnan(a_res[i])
                continue
            assert a_res[i] == pytest.approx(a1, rel=1e-12)
            np.testing.assert_allclose(b_res[i], np.ravel(b1), rtol=1e-10, atol=1e-10 * np.abs(b1).max())
        assert np.isnan(a_res[B - 1]) and np.isnan(a_res[11])   # NaN and inf poison their own instance, as in the reference
        clean = [i for i in range(B) if i not in flagged]
        assert np.all(np.isfinite(a_res[clean])) and np.all(np.isfinite(b_res[clean]))
        a_ref, b_ref = pwl_reference(kind, direction, az[clean[:6]], bz[clean[:6]], ax[clean[:6]], bx[clean[:6]])
        for k, i in enumerate(clean[:6]):
            close_a(a_res[i], a_ref[k], (ax if direction == "fwd" else az)[i], dtype)
            close(b_res[i], b_ref[k], dtype, scale=10 * np.abs(b_ref[k]).max())


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_pwl_resident_damping_and_aliased_output(T, ctx, dtype):
    """damping against the buffer being overwritten (what the EP scheduler does) runs on the pipeline; an update written
    over its own input cannot be redone from its inputs, so it takes the tile kernels - same numbers either way"""
    B, n = shape_for(ctx, dtype)
    rs = np.random.RandomState(8)
    d = 0.25
    az, ax = rs.uniform(0.8, 2.0, B), rs.uniform(0.5, 1.5, B)
    bz = rs.standard_normal((B, n)).astype(dtype); bx = rs.standard_normal((B, n)).astype(dtype)
    old_a = rs.uniform(0.5, 2.0, B); old_b = rs.standard_normal((B, n)).astype(dtype)
    ch = T.ReluChannel(dtype=dtype, ctx=ctx)
    oa = ctx.scalars(old_a); ob = ctx.array(old_b, dtype)
    runs0 = ctx.resident_runs()
    ch.compute_backward_message(ctx.scalars(az), ctx.array(bz, dtype), ctx.scalars(ax), ctx.array(bx, dtype), out_a=oa, out_b=ob,
                                damping=d, old_a=oa, old_b=ob)
    assert ctx.resident_runs() == runs0 + 1
    a_ref, b_ref = pwl_reference("relu", "bwd", az, bz, ax, bx)
    a_new, b_new = oa.get(), ob.get()
    for i in range(0, B, 7):
        close_a(a_new[i], d * old_a[i] + (1 - d) * a_ref[i], az[i], dtype)
        close(b_new[i], d * old_b[i].astype(np.float64) + (1 - d) * b_ref[i], dtype, scale=10 * max(np.abs(b_ref[i]).max(), 1.0))
    bio = ctx.array(bz, dtype); ao = ctx.empty((B,), np.float64)
    runs0 = ctx.resident_runs()
    ch.compute_backward_message(ctx.scalars(az), bio, ctx.scalars(ax), ctx.array(bx, dtype), out_a=ao, out_b=bio)
    assert ctx.resident_runs() == runs0, "an in-place update must not take the lean pipeline (it could not be redone)"
    b_new = bio.get()
    for i in range(0, B, 7):
        close(b_new[i], b_ref[i], dtype, scale=10 * max(np.abs(b_ref[i]).max(), 1.0))


@pytest.mark.parametrize("which", ["relu", "gb"])
def test_resident_gives_up_cleanly(which):
    """A hand-off that never arrives (forced: one tile's sum is withheld, deadline 150 ms) must not trap.  The kernel
    drains with NaN for the instances it could not finish and raises the host-visible error word; the context takes the
    tile kernels from then on.  For the lean K7 form the NaN instances are redone by the tile kernels in the same call:
    no NaN reaches the caller and the redone instances equal a pure tile-kernel run bit for bit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TRAMP_EW_RESIDENT_TIMEOUT_MS="150", TRAMP_EW_RESIDENT_DBG_SKIP="19")    # tile 4 of instance 2
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_resident_giveup_worker.py"), which], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]                # (a trap would have killed the process)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["resident_ran"] and out["gave_up"] and out["second_call_on_tiles"], out
    assert out["nan_instances_second"] == 0 and out["oracle_instance0"] and out["finite_agree"], out
    if which == "relu":
        assert out["nan_instances_first"] == 0, out                       # redone in the same call
        assert out["bitwise_equal_instances"] >= 1, out                   # the withheld instance: computed by the tile kernels both times
    else:
        assert 1 <= out["nan_instances_first"] < out["B"], out            # loud, and confined to the instances not finished
