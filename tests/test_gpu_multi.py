"""GPU tests of the in-process multi-GPU context (sgp_ctx_create_multi, stheno.jl_amd/csrc/multi.hip):
one `sgp_logpdf` call sharded over several ranks.  The GPU box has ONE MI355X, so
  * several ranks on device 0 ("loopback" transport: same-device copies) exercise the whole multi-rank
    orchestration -- packed block-cyclic panels, look-ahead, double-buffered receives, reductions --
    with the real kernels, and
  * a one-rank context exercises the RCCL path (dlopen'ed librccl: ncclCommInitAll, grouped
    ncclBroadcast per panel, ncclAllReduce of the scalars) exactly as an 8-GPU context issues it.
Reference values: the single-GPU driver and the CPU oracle."""
import os

import numpy as np
import pytest

import stheno_jl_amd as P
from oracle import reference_model as orm

pytestmark = pytest.mark.gpu


def _problem(N, D=3, seed=123456):
    rng = np.random.default_rng(seed)
    F = P.gppp_sum_model()
    n1 = N // 3
    xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (n1, n1, N - 2 * n1)]
    x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
    y = rng.standard_normal(N)
    return F, x, xs, y


def _with_ctx(ctx, fn):
    prev = P.lib.set_default_context(ctx)
    try:
        return fn()
    finally:
        P.lib.set_default_context(prev)


@pytest.fixture
def panel128(monkeypatch):
    monkeypatch.setenv("SGP_MULTI_PANEL", "128")


@pytest.mark.parametrize("nranks", [1, 2, 3, 5])
def test_loopback_ranks_match_single_gpu_and_oracle(panel128, nranks):
    ctx = P.lib.Context(devices=[0] * nranks) if nranks > 1 else None
    for N in (300, 1000, 1411):
        F, x, xs, y = _problem(N)
        v0 = P.logpdf(F(x, 0.1), y)
        ref = orm.gppp_sum_logpdf(xs, y, 0.1)
        assert abs(v0 - ref) <= 1e-10 * abs(ref)
        if ctx is None:
            continue
        assert ctx.ndev == nranks and ctx.transport == "loopback"
        v = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
        assert abs(v - ref) <= 1e-10 * abs(ref), (nranks, N, v, ref)
        # diagonal noise and a prior mean go through the sharded path as well
        noise = 0.05 + np.random.default_rng(N).random(N)
        v_d = _with_ctx(ctx, lambda: P.logpdf(F(x, noise), y))
        assert abs(v_d - P.logpdf(F(x, noise), y)) <= 1e-11 * abs(v_d)
    if ctx is not None:
        ctx.close()


def test_larger_problem_default_panel_width(monkeypatch):
    monkeypatch.delenv("SGP_MULTI_PANEL", raising=False)
    ctx = P.lib.Context(devices=[0, 0])
    F, x, xs, y = _problem(5000, D=4)
    v0 = P.logpdf(F(x, 0.1), y)
    v = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    assert abs(v - v0) <= 1e-11 * abs(v0)
    # repeated calls reuse the ranks' storage
    v2 = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    assert v2 == v
    ctx.close()


def test_posdef_failure_is_reported_from_any_rank(panel128):
    ctx = P.lib.Context(devices=[0, 0, 0])
    F, x, xs, y = _problem(700)
    with pytest.raises(P.PosDefException) as e0:
        P.logpdf(F(x, -5.0), y)
    with pytest.raises(P.PosDefException) as e1:
        _with_ctx(ctx, lambda: P.logpdf(F(x, -5.0), y))
    assert e1.value.info == e0.value.info
    ctx.close()


def test_rccl_transport_one_rank(panel128, monkeypatch):
    """devices = [0]: distinct devices -> transport auto = RCCL.  Every collective of the 8-GPU path is
    issued (grouped broadcast per panel, all-reduce of the scalars) on a one-rank communicator."""
    monkeypatch.setenv("SGP_MULTI_TRANSPORT", "rccl")
    ctx = P.lib.Context(devices=[0])
    assert ctx.ndev == 1 and ctx.transport == "rccl"
    for N in (500, 2000):
        F, x, xs, y = _problem(N)
        ref = orm.gppp_sum_logpdf(xs, y, 0.1)
        v = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
        assert abs(v - ref) <= 1e-10 * abs(ref)
    ctx.close()


def test_p2p_transport_single_device_is_loopback_free(monkeypatch):
    monkeypatch.setenv("SGP_MULTI_TRANSPORT", "p2p")
    ctx = P.lib.Context(devices=[0])
    assert ctx.transport == "p2p"
    F, x, xs, y = _problem(600)
    v = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    assert abs(v - P.logpdf(F(x, 0.1), y)) <= 1e-12 * abs(v)
    ctx.close()


# ---- round 3: every Cholesky-based operator behind the multi-GPU context ---------------------------------
def _post_problem(N, ns, D=3, seed=7):
    F, x, xs, y = _problem(N, D=D, seed=seed)
    rng = np.random.default_rng(seed + 1)
    xnew = P.BlockData([P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((D, ns))))),
                        P.GPPPInput("f1", P.ColVecs(np.asfortranarray(rng.standard_normal((D, 17)))))])
    return F, x, xs, y, xnew


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_posterior_on_kept_sharded_factor(panel128, nranks):
    """posterior(fx, y) on a multi-GPU context keeps the sharded factor; mean / var / cov at x* and alpha against the
    single-GPU driver (which the parity suite holds against the oracle), repeated predictions on one posterior."""
    ctx = P.lib.Context(devices=[0] * nranks)
    for N, ns in ((300, 40), (1411, 150)):
        F, x, xs, y, xnew = _post_problem(N, ns)
        noise = 0.05 + np.random.default_rng(N).random(N)
        for nz in (0.1, noise):
            p0 = P.posterior(F(x, nz), y)
            m0, c0 = p0.mean_and_cov(xnew)
            v0 = p0.var(xnew)
            pm = _with_ctx(ctx, lambda: P.posterior(F(x, nz), y))
            assert np.max(np.abs(pm.alpha - p0.alpha)) <= 1e-9 * np.max(np.abs(p0.alpha))
            for _ in range(2):      # the factor stays resident: any number of predictions
                m1, c1 = pm.mean_and_cov(xnew)
                v1 = pm.var(xnew)
                assert np.max(np.abs(m1 - m0)) <= 1e-10 * max(1.0, np.max(np.abs(m0)))
                assert np.max(np.abs(v1 - v0)) <= 1e-10
                assert np.max(np.abs(c1 - c0)) <= 1e-10
            # a second, different query against the same factor
            x2 = P.GPPPInput("f2", P.ColVecs(np.asfortranarray(np.random.default_rng(3).standard_normal((3, 33)))))
            ma, va = pm.mean_and_var(x2)
            mb, vb = p0.mean_and_var(x2)
            assert np.max(np.abs(ma - mb)) <= 1e-10 * max(1.0, np.max(np.abs(mb))) and np.max(np.abs(va - vb)) <= 1e-10
    ctx.close()


def test_posterior_multi_against_oracle(panel128):
    ctx = P.lib.Context(devices=[0, 0, 0])
    rng = np.random.default_rng(11)
    D, n = 4, 257
    F = P.gppp_sum_model()
    xs = [np.asfortranarray(rng.standard_normal((D, n + k))) for k in range(3)]
    x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
    y = rng.standard_normal(len(x))
    xnew = P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((D, 33)))))
    m, v = _with_ctx(ctx, lambda: P.mean_and_var(P.posterior(F(x, 0.1), y)(xnew)))
    m_ref, v_ref = orm.gppp_sum_posterior(xs, y, 0.1, xnew.x.X)
    assert np.max(np.abs(m - m_ref)) < 1e-8 * max(1.0, np.max(np.abs(m_ref)))
    assert np.max(np.abs(v - (v_ref + 1e-18))) < 1e-8 * max(1.0, np.max(np.abs(v_ref)))
    ctx.close()


@pytest.mark.parametrize("nranks", [2, 3])
def test_logpdf_matrix_rhs_and_rand_multi(panel128, nranks):
    ctx = P.lib.Context(devices=[0] * nranks)
    F, x, xs, y = _problem(900)
    Y = np.asfortranarray(np.random.default_rng(5).standard_normal((900, 3)))
    v0 = P.logpdf(F(x, 0.1), Y)
    v1 = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), Y))
    assert v1.shape == (3,) and np.max(np.abs(v1 - v0)) <= 1e-11 * np.max(np.abs(v0))
    Z = np.asfortranarray(np.random.default_rng(6).standard_normal((900, 5)))
    r0 = P.rand(None, F(x, 0.1), 5, Z=Z)
    r1 = _with_ctx(ctx, lambda: P.rand(None, F(x, 0.1), 5, Z=Z))
    r2 = _with_ctx(ctx, lambda: P.rand(None, F(x, 0.1), 5, Z=Z))
    assert np.max(np.abs(r1 - r0)) <= 1e-11 * np.max(np.abs(r0))
    assert np.array_equal(r1, r2)          # deterministic: fixed-order reduction over the ranks
    ctx.close()


@pytest.mark.parametrize("nranks", [2, 3])
def test_elbo_data_sharded_multi(nranks):
    """elbo on a multi-GPU context: data points sharded over the ranks (one host thread each), one reduction."""
    ctx = P.lib.Context(devices=[0] * nranks)
    rng = np.random.default_rng(21)
    D, N, M = 3, 1500, 96
    f = P.stretch(P.atomic(P.GP(P.SEKernel()), P.GPC()), 0.7)
    X = np.asfortranarray(rng.standard_normal((D, N)))
    Z = np.asfortranarray(X[:, :M] + 0.01)
    y = rng.standard_normal(N)
    for nz in (0.1, 0.05 + rng.random(N)):
        fx, fz = f(P.ColVecs(X), nz), f(P.ColVecs(Z), 1e-6)
        e0 = P.elbo(P.VFE(fz), fx, y)
        e1 = _with_ctx(ctx, lambda: P.elbo(P.VFE(fz), fx, y))
        assert abs(e1 - e0) <= 1e-10 * abs(e0), (e0, e1)
    # fewer data points than ranks would leave empty slices: still exact
    Xs = np.asfortranarray(X[:, :nranks - 1])
    fx, fz = f(P.ColVecs(Xs), 0.1), f(P.ColVecs(Z[:, :8]), 1e-6)
    e0 = P.elbo(P.VFE(fz), fx, y[:nranks - 1])
    e1 = _with_ctx(ctx, lambda: P.elbo(P.VFE(fz), fx, y[:nranks - 1]))
    assert abs(e1 - e0) <= 1e-10 * abs(e0)
    ctx.close()


def test_sparse_posterior_data_sharded_multi():
    """posterior(VFE(fz), fx, y) on a multi-GPU context: the N-sized work is sharded like the ELBO's, the M x M
    factors are kept on devices[0] and answer predictions as on one GPU."""
    ctx = P.lib.Context(devices=[0, 0, 0])
    rng = np.random.default_rng(22)
    D, N, M = 2, 1100, 64
    f = P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 0.8)
    X = np.asfortranarray(rng.standard_normal((D, N)))
    Z = np.asfortranarray(rng.standard_normal((D, M)))
    y = rng.standard_normal(N)
    xs = P.ColVecs(np.asfortranarray(rng.standard_normal((D, 40))))
    fx, fz = f(P.ColVecs(X), 0.2), f(P.ColVecs(Z), 1e-6)
    q0 = P.posterior(P.VFE(fz), fx, y)
    q1 = _with_ctx(ctx, lambda: P.posterior(P.VFE(fz), fx, y))
    m0, c0 = q0.mean_and_cov(xs)
    m1, c1 = q1.mean_and_cov(xs)
    assert np.max(np.abs(m1 - m0)) <= 1e-9 * max(1.0, np.max(np.abs(m0)))
    assert np.max(np.abs(c1 - c0)) <= 1e-9
    assert np.max(np.abs(q1.var(xs) - q0.var(xs))) <= 1e-9
    del q1
    ctx.close()


def test_broadcast_forms_agree_bit_for_bit(panel128, monkeypatch):
    """scatter + all-gather (default for >= 3 ranks) vs one copy owner -> receiver: the same panels arrive."""
    F, x, xs, y = _problem(1411)
    vals = []
    for form in ("allgather", "direct"):
        monkeypatch.setenv("SGP_MULTI_BCAST", form)
        ctx = P.lib.Context(devices=[0] * 5)
        vals.append(_with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y)))
        ctx.close()
    assert vals[0] == vals[1]


def test_multi_stats_and_profile(panel128):
    import ctypes as C
    ctx = P.lib.Context(devices=[0, 0, 0])
    F, x, xs, y = _problem(1000)
    _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    out = np.zeros(64)
    n = C.c_int64()
    P.lib.check(ctx.lib.sgp_ctx_multi_stats(ctx.handle, P.lib.dptr(out), 64, C.byref(n)))
    assert n.value == 11 + 4 * 3 and out[0] == 3 and out[1] > 0 and out[5] == 8
    assert out[9 + 4 * 3] in (0.0, 1.0) and out[10 + 4 * 3] >= 0   # ownership of the last geometry; late-bound event waits
    assert 0 < out[8 + 4 * 3] <= out[1]                            # host enqueue time of that call <= its wall time
    assert all(out[8 + 4 * i] > 0 for i in range(3))              # every rank did trailing updates
    assert sum(out[10 + 4 * i] for i in range(3)) == 8             # the 8 panels were factored exactly once
    P.lib.check(ctx.lib.sgp_ctx_multi_profile(ctx.handle, 1))
    v = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    assert abs(v - P.logpdf(F(x, 0.1), y)) <= 1e-11 * abs(v)
    P.lib.check(ctx.lib.sgp_ctx_multi_profile_get(ctx.handle, None, 0, C.byref(n)))
    assert n.value == 8 * (3 + 3 * 3)     # per panel: factor, look-ahead update, bytes, then near A / near B / far per rank
    prof = np.zeros(n.value)
    P.lib.check(ctx.lib.sgp_ctx_multi_profile_get(ctx.handle, P.lib.dptr(prof), n.value, C.byref(n)))
    prof = prof.reshape(8, 12)
    assert np.all(prof[:, 0] > 0) and np.all(prof[:, 2] > 0)
    assert out[7] == 1                    # panels per update group (SGP_MULTI_GROUP, default 1)
    P.lib.check(ctx.lib.sgp_ctx_multi_profile(ctx.handle, 0))
    ctx.close()


# ---- round 4: grouped (deep-K) far updates, mixed panel widths, the batched update entry point -------------------------
@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_update_groups_give_the_same_bits_and_panel_width_mixtures_the_same_value(monkeypatch, nranks):
    """SGP_MULTI_GROUP (panels applied per far update: K = G panels) only regroups the trailing updates: every tile still
    sees k ascending through the same tile program, so for ONE panel layout logpdf comes out bit for bit the same whatever
    G.  SGP_MULTI_PANEL / _TAIL / _TAIL_FRAC (mixed panel widths) change which columns a panel's log det and |z|^2 partial
    sums cover (their order of summation), so across layouts the value agrees to rounding -- and with the single-GPU driver
    and the oracle to 1e-10."""
    F, x, xs, y = _problem(2900, D=3)
    ref = orm.gppp_sum_logpdf(xs, y, 0.1)
    single = P.logpdf(F(x, 0.1), y)
    layouts = [{"SGP_MULTI_PANEL": "128"},
               {"SGP_MULTI_PANEL": "256", "SGP_MULTI_PANEL_TAIL": "128"},
               {"SGP_MULTI_PANEL": "512", "SGP_MULTI_PANEL_TAIL": "128", "SGP_MULTI_TAIL_FRAC": "0.4"},
               {"SGP_MULTI_PANEL": "384", "SGP_MULTI_PANEL_TAIL": "256", "SGP_MULTI_TAIL_FRAC": "0.0"}]
    for lay in layouts:
        vals = []
        for G in ("1", "2", "3", "4", "8"):
            for k in ("SGP_MULTI_PANEL", "SGP_MULTI_PANEL_TAIL", "SGP_MULTI_TAIL_FRAC", "SGP_MULTI_GROUP"):
                monkeypatch.delenv(k, raising=False)
            for k, v in lay.items():
                monkeypatch.setenv(k, v)
            monkeypatch.setenv("SGP_MULTI_GROUP", G)
            ctx = P.lib.Context(devices=[0] * nranks)
            for rep in range(2):     # second call: ring buffers and stores reused
                vals.append(_with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y)))
            Ym = np.column_stack([y, 2.0 * y - 1.0, np.cos(y)])
            vm = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), Ym))
            assert vm[0] == vals[-1]
            ctx.close()
        assert all(v == vals[0] for v in vals), (lay, vals)
        assert abs(vals[0] - ref) <= 1e-10 * abs(ref)
        assert abs(vals[0] - single) <= 1e-12 * abs(ref)


def test_posterior_and_rand_on_mixed_width_grouped_factor(monkeypatch):
    monkeypatch.setenv("SGP_MULTI_PANEL", "256")
    monkeypatch.setenv("SGP_MULTI_PANEL_TAIL", "128")
    monkeypatch.setenv("SGP_MULTI_GROUP", "4")
    F, x, xs, y, xnew = _post_problem(1700, 40)
    p0 = P.posterior(F(x, 0.1), y)
    m0, v0 = P.mean_and_var(p0(xnew))
    ctx = P.lib.Context(devices=[0] * 3)
    pm = _with_ctx(ctx, lambda: P.posterior(F(x, 0.1), y))
    m1, v1 = _with_ctx(ctx, lambda: P.mean_and_var(pm(xnew)))
    assert np.abs(m1 - m0).max() <= 1e-9 * max(1.0, np.abs(m0).max())
    assert np.abs(v1 - v0).max() <= 1e-9 * max(1.0, np.abs(v0).max())
    Z = np.random.default_rng(3).standard_normal((len(x), 4))
    r0 = P.rand(None, F(x, 0.1), 4, Z=Z)
    r1 = _with_ctx(ctx, lambda: P.rand(None, F(x, 0.1), 4, Z=Z))
    assert np.abs(r1 - r0).max() <= 1e-9 * np.abs(r0).max()
    del pm
    ctx.close()


def test_batched_panel_update_entry_point_is_bit_identical_to_single_updates():
    """sgp_dev_panel_update_batch (one launch: several packed destination panels, each contracted over a range of packed
    source panels) against the same updates issued one sgp_dev_panel_update at a time."""
    import ctypes as C
    import torch
    lib = P.lib.load()
    ctx = P.lib.default_context()
    rng = np.random.default_rng(11)
    m_tot = 128 * 13
    srcs = [(0, 256), (256, 128), (384, 384)]                 # (first column = first stored row, width) of three factored panels
    dsts = [(768, 256, 0, 3), (1024, 128, 1, 2), (1152, 384, 0, 1), (1536, 128, 2, 1)]   # (c0, w, src_first, src_count)
    dev = torch.device("cuda", ctx.device)
    # packed column-major panels as flat buffers: element (r, k) at (r - row0) + k * ld
    sbuf = [torch.from_numpy(np.asfortranarray(rng.standard_normal((m_tot - r0, w))).ravel(order="F").copy()).to(dev)
            for r0, w in srcs]
    dbuf0 = [torch.from_numpy(np.asfortranarray(rng.standard_normal((m_tot - c0, w))).ravel(order="F").copy()).to(dev)
             for c0, w, _, _ in dsts]
    one = [t.clone() for t in dbuf0]
    bat = [t.clone() for t in dbuf0]
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        h = stream.cuda_stream
        for d, (c0, w, s0, sn) in enumerate(dsts):
            for q in range(s0, s0 + sn):
                r0, sw = srcs[q]
                P.lib.check(lib.sgp_dev_panel_update(ctx.handle, sbuf[q].data_ptr(), m_tot - r0, r0, sw,
                                                     one[d].data_ptr() - 8 * c0, m_tot - c0, c0, w, m_tot, h))
        src_arr = (P.lib.sgp_panel_src * len(srcs))()
        for q, (r0, sw) in enumerate(srcs):
            src_arr[q] = P.lib.sgp_panel_src(sbuf[q].data_ptr(), m_tot - r0, r0, sw)
        dst_arr = (P.lib.sgp_panel_dst * len(dsts))()
        for d, (c0, w, s0, sn) in enumerate(dsts):
            dst_arr[d] = P.lib.sgp_panel_dst(bat[d].data_ptr(), m_tot - c0, c0, w, s0, sn)
        P.lib.check(lib.sgp_dev_panel_update_batch(ctx.handle, src_arr, len(srcs), dst_arr, len(dsts), m_tot, h))
    stream.synchronize()
    for d, (c0, w, _, _) in enumerate(dsts):
        a = one[d].cpu().numpy().reshape((m_tot - c0, w), order="F")
        b = bat[d].cpu().numpy().reshape((m_tot - c0, w), order="F")
        orig = dbuf0[d].cpu().numpy().reshape((m_tot - c0, w), order="F")
        # lower trapezoid (tile rows >= tile columns) updated, identical bits; the strictly upper tiles untouched
        low = (np.arange(m_tot - c0)[:, None] // 128) >= (np.arange(w)[None, :] // 128)
        assert np.array_equal(a[low], b[low])
        assert np.array_equal(b[~low], orig[~low])
        assert not np.array_equal(b[low], orig[low])
    # and against NumPy for one destination
    c0, w, s0, sn = dsts[0]
    want = dbuf0[0].cpu().numpy().reshape((m_tot - c0, w), order="F").copy()
    for q in range(s0, s0 + sn):
        r0, sw = srcs[q]
        Pq = sbuf[q].cpu().numpy().reshape((m_tot - r0, sw), order="F")
        want -= Pq[c0 - r0:, :] @ Pq[c0 - r0:c0 - r0 + w, :].T
    got = bat[0].cpu().numpy().reshape((m_tot - c0, w), order="F")
    low = (np.arange(m_tot - c0)[:, None] // 128) >= (np.arange(w)[None, :] // 128)
    assert np.abs(got[low] - want[low]).max() <= 1e-11 * np.abs(want).max()


def test_predicting_on_a_posterior_whose_context_is_gone_fails_cleanly(panel128):
    """Advisor, round 3: a posterior handle may outlive its context (it is freed without it), but predicting needs the
    context's ranks, streams and scratch -- that was a use-after-free; now an error, for the single-GPU and the sharded
    posterior alike."""
    F, x, xs, y, xnew = _post_problem(700, 20)
    for devices in ([0], [0, 0, 0]):
        ctx = P.lib.Context(devices=devices) if len(devices) > 1 else P.lib.Context(0)
        post = _with_ctx(ctx, lambda: P.posterior(F(x, 0.1), y))
        m0 = _with_ctx(ctx, lambda: post.mean(xnew))
        assert np.all(np.isfinite(m0))
        ctx.close()
        with pytest.raises(P.SthenoMIError) as e:
            post.mean(xnew)
        assert "destroyed" in str(e.value)
        del post


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_dense_observation_noise_on_the_multi_gpu_context(panel128, nranks):
    """Round 4: f(x, S::Matrix) -- a dense Sigma_y (/root/reference/test/affine_transformations/test_util.jl:114-120 treats
    dense, diagonal and isotropic noise alike) -- no longer falls back to devices[0]: the owner of a panel adds its column
    slab of Sigma_y at assembly, nothing travels.  logpdf, posterior and rand against the single-GPU driver and the oracle."""
    import oracle.abstractgps as oagp
    import oracle.kernelfunctions as okf
    import oracle.stheno as ost
    import models
    N = 1111
    F, x, xs, y, xnew = _post_problem(N, 25)
    rng = np.random.default_rng(5)
    B = rng.standard_normal((N, 7))
    S = 0.05 * np.eye(N) + 0.01 * B @ B.T                      # dense, positive definite
    v0 = P.logpdf(F(x, S), y)
    fo, go = models.gppp_docstring(models.oracle_api())
    Fo = ost.GPPP(fo, go)
    xo = ost.BlockData([ost.GPPPInput(k, okf.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
    ref = oagp.logpdf(Fo(xo, S), y)
    assert abs(v0 - ref) <= 1e-10 * abs(ref)
    ctx = P.lib.Context(devices=[0] * nranks)
    v = _with_ctx(ctx, lambda: P.logpdf(F(x, S), y))
    assert abs(v - ref) <= 1e-10 * abs(ref) and abs(v - v0) <= 1e-12 * abs(ref)
    p0 = P.posterior(F(x, S), y)
    pm = _with_ctx(ctx, lambda: P.posterior(F(x, S), y))
    m0, c0 = P.mean_and_cov(p0(xnew))
    m1, c1 = _with_ctx(ctx, lambda: P.mean_and_cov(pm(xnew)))
    assert np.abs(m1 - m0).max() <= 1e-9 * max(1.0, np.abs(m0).max())
    assert np.abs(c1 - c0).max() <= 1e-9 * max(1.0, np.abs(c0).max())
    Z = np.random.default_rng(9).standard_normal((N, 3))
    r0 = P.rand(None, F(x, S), 3, Z=Z)
    r1 = _with_ctx(ctx, lambda: P.rand(None, F(x, S), 3, Z=Z))
    assert np.abs(r1 - r0).max() <= 1e-9 * np.abs(r0).max()
    del pm
    ctx.close()


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_logpdf_gradient_sharded_over_the_ranks(panel128, nranks):
    """Round 4: sgp_logpdf_grad on a multi-GPU context -- the kept sharded factor, L^-T through the posterior's row sweep,
    C^-1 = sum over ranks of X_i X_i' with a reduce-scatter by column slabs, every rank contracting its slabs with the
    kernel derivatives -- against the single-GPU gradient (scalar and diagonal noise, a prior mean, several blocks and
    terms; what Zygote derives for examples/getting_started/script.jl:154-213)."""
    F, x, xs, y = _problem(1411, D=3)
    ctx = P.lib.Context(devices=[0] * nranks)
    for noise in (0.1, 0.05 + np.random.default_rng(3).random(len(y))):
        g0 = P.logpdf_and_gradient(F(x, noise), y)
        g1 = _with_ctx(ctx, lambda: P.logpdf_and_gradient(F(x, noise), y))
        assert abs(g1["logpdf"] - g0["logpdf"]) <= 1e-11 * abs(g0["logpdf"])
        assert np.abs(g1["y"] - g0["y"]).max() <= 1e-9 * np.abs(g0["y"]).max()
        assert np.abs(g1["mean"] - g0["mean"]).max() <= 1e-9 * np.abs(g0["mean"]).max()
        assert np.abs(np.asarray(g1["noise"]) - np.asarray(g0["noise"])).max() <= 1e-8 * max(1.0, np.abs(np.asarray(g0["noise"])).max())
        assert len(g1["terms"]) == len(g0["terms"]) > 1
        for t0, t1 in zip(g0["terms"], g1["terms"]):
            assert (t0["I"], t0["J"], t0["kind"]) == (t1["I"], t1["J"], t1["kind"])
            assert abs(t1["d_coef"] - t0["d_coef"]) <= 1e-8 * max(1.0, abs(t0["d_coef"]))
            assert abs(t1["d_inscale"] - t0["d_inscale"]) <= 1e-8 * max(1.0, abs(t0["d_inscale"]))
    # and against central differences of the sharded logpdf itself, for the noise
    h = 1e-5
    fd = (_with_ctx(ctx, lambda: P.logpdf(F(x, 0.1 + h), y)) - _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1 - h), y))) / (2 * h)
    g = _with_ctx(ctx, lambda: P.logpdf_and_gradient(F(x, 0.1), y))
    assert abs(g["noise"] - fd) <= 1e-5 * max(1.0, abs(fd))
    ctx.close()


@pytest.mark.parametrize("nranks", [2, 5, 8])
def test_per_rank_enqueue_threads_give_the_same_bits_as_one_thread(monkeypatch, nranks):
    """Round 4: the sweep of the sharded factorisation is issued by one enqueue thread per rank (SGP_MULTI_THREADS, default on
    with more than one rank): every thread walks the same schedule and issues its own rank's calls, cross-rank events are
    ordered through per-event sequence numbers.  Same bits as the one-thread enqueue, for both panel transports' copy forms,
    sub-panel pipelining on and off, repeated calls."""
    F, x, xs, y = _problem(2500, D=3)
    ref = orm.gppp_sum_logpdf(xs, y, 0.1)
    for env in ({"SGP_MULTI_PANEL": "128"}, {"SGP_MULTI_PANEL": "512", "SGP_MULTI_SUBPANEL": "128"},
                {"SGP_MULTI_PANEL": "256", "SGP_MULTI_SUBPANEL": "128", "SGP_MULTI_BCAST": "direct", "SGP_MULTI_GROUP": "2"}):
        vals = []
        for threads in ("0", "1"):
            for k in ("SGP_MULTI_PANEL", "SGP_MULTI_SUBPANEL", "SGP_MULTI_BCAST", "SGP_MULTI_GROUP"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            monkeypatch.setenv("SGP_MULTI_THREADS", threads)
            ctx = P.lib.Context(devices=[0] * nranks)
            for rep in range(3):
                vals.append(_with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y)))
            ctx.close()
        assert all(v == vals[0] for v in vals), (env, vals)
        assert abs(vals[0] - ref) <= 1e-10 * abs(ref)


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_structural_zeros_on_the_sharded_factorisation(monkeypatch, nranks):
    """Round 4: the sum model's covariance has an exact zero block (f1 and f2 are independent) and so has its factor; the
    sharded factorisation skips, tile by tile, the source panels whose k tiles are structurally zero for it
    (gemm_nt.hip: gemm_nt_seg_kernel; the pattern comes from rank 0's context, tests/test_gpu_struct_zeros.py has the
    single-GPU side).  Same bits with the skipping on and off -- logpdf, kept-factor posterior, draws -- for whole panels,
    sub-panel pipelining and panel groups; the work counter shows the skipped share."""
    F, x, xs, y = _problem(3000, D=3)
    ref = orm.gppp_sum_logpdf(xs, y, 0.1)
    rng = np.random.default_rng(1)
    xs_new = P.BlockData([P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((3, 30)))))])
    Z = np.asfortranarray(rng.standard_normal((3000, 2)))

    def run():
        fx = F(x, 0.1)
        post = P.posterior(fx, y)
        m, v = post.mean_and_var(xs_new)
        return dict(lp=np.array([P.logpdf(fx, y)]), m=np.asarray(m), v=np.asarray(v), r=np.asarray(P.rand(None, fx, 2, Z=Z)))

    for env in ({"SGP_MULTI_PANEL": "128"}, {"SGP_MULTI_PANEL": "512", "SGP_MULTI_SUBPANEL": "128"},
                {"SGP_MULTI_PANEL": "256", "SGP_MULTI_SUBPANEL": "0", "SGP_MULTI_GROUP": "2"}):
        outs = []
        for sz in ("0", "1"):
            for k in ("SGP_MULTI_PANEL", "SGP_MULTI_SUBPANEL", "SGP_MULTI_GROUP"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            monkeypatch.setenv("SGP_STRUCT_ZEROS", sz)
            # (one ownership for both runs: the balanced table of round 5 is built from the pattern, so it differs between
            # skipping on and off, and posterior / rand add per-rank partial sums in rank order)
            monkeypatch.setenv("SGP_MULTI_OWNERS", "cyclic")
            ctx = P.lib.Context(devices=[0] * nranks)
            outs.append(_with_ctx(ctx, run))
            _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
            e, d = ctx.factor_work()
            assert (e < 0.75 * d) if sz == "1" else (e == d), (env, sz, e, d)
            ctx.close()
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[1][k]), (env, k)
        assert abs(outs[0]["lp"][0] - ref) <= 1e-10 * abs(ref)


# ---- round 5: who owns which panel --------------------------------------------------------------------------------------
def _stats(ctx, nranks):
    import ctypes as C
    out = np.zeros(11 + 4 * nranks)
    n = C.c_int64()
    P.lib.check(ctx.lib.sgp_ctx_multi_stats(ctx.handle, P.lib.dptr(out), len(out), C.byref(n)))
    return out


@pytest.mark.parametrize("nranks", [2, 3, 5, 8])
def test_any_ownership_table_gives_the_cyclic_deals_bits(monkeypatch, nranks):
    """Round 5: the panels of the sharded factorisation are dealt out by a table (own_table.h; balanced from the symbolic tile
    pattern of a structured model, SGP_MULTI_OWNERS = cyclic | balanced | an explicit list).  The factor does not depend on
    who owns a panel -- every tile sees the same products in the same order -- and logdet / the quadratic forms are added
    per panel in panel order, so logpdf comes out bit for bit the same for every table: cyclic, balanced, two scrambled
    lists (one of them with a rank owning consecutive panels and a rank owning none), vector and matrix right-hand sides,
    whole panels and the sub-panel pipeline, skipping on.  Posterior moments and draws add per-rank partial sums: same values
    to rounding, checked against the cyclic deal and the oracle."""
    F, x, xs, y = _problem(3100, D=3)
    ref = orm.gppp_sum_logpdf(xs, y, 0.1)
    rng = np.random.default_rng(5)
    Y = np.asfortranarray(rng.standard_normal((3100, 3)))
    xs_new = P.BlockData([P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((3, 21)))))])
    Z = np.asfortranarray(rng.standard_normal((3100, 2)))
    scr1 = ",".join(str((7 * j + 3) % nranks) for j in range(11))
    scr2 = ",".join(str(v) for v in ([0, 0, 0] + [nranks - 1] * 2 + list(range(max(1, nranks - 1)))))   # consecutive panels on one rank
    for env in ({"SGP_MULTI_PANEL": "128"}, {"SGP_MULTI_PANEL": "512", "SGP_MULTI_SUBPANEL": "128"}):
        vals, modes, posts = [], [], []
        for owners in ("cyclic", "balanced", scr1, scr2):
            for k in ("SGP_MULTI_PANEL", "SGP_MULTI_SUBPANEL"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            monkeypatch.setenv("SGP_MULTI_OWNERS", owners)
            ctx = P.lib.Context(devices=[0] * nranks)

            def run():
                fx = F(x, 0.1)
                lp = P.logpdf(fx, y)
                mode = _stats(ctx, nranks)[9 + 4 * nranks]
                lpY = np.asarray(P.logpdf(fx, Y))
                m, v = P.posterior(fx, y).mean_and_var(xs_new)
                return lp, mode, lpY, np.asarray(m), np.asarray(v), np.asarray(P.rand(None, fx, 2, Z=Z))

            lp, mode, lpY, m, v, r = _with_ctx(ctx, run)
            ctx.close()
            vals.append((lp, lpY))
            modes.append(mode)
            posts.append((m, v, r))
        assert modes[0] == 0 and modes[2] == 2 and modes[3] == 2, modes
        assert modes[1] in (0.0, 1.0)
        if env["SGP_MULTI_PANEL"] == "128" and nranks >= 3:
            assert modes[1] == 1.0   # 25 panels of a model with a zero block: the table is built (and differs from the deal)
        for lp, lpY in vals[1:]:
            assert lp == vals[0][0] and np.array_equal(lpY, vals[0][1]), (env, vals)
        assert abs(vals[0][0] - ref) <= 1e-10 * abs(ref)
        for m, v, r in posts[1:]:
            np.testing.assert_allclose(m, posts[0][0], rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(v, posts[0][1], rtol=1e-9, atol=1e-11)
            np.testing.assert_allclose(r, posts[0][2], rtol=1e-9, atol=1e-10)


def test_balanced_table_is_what_a_structured_model_gets_by_default(monkeypatch):
    """The default ownership of a structured model is the balanced table (stats slot 9 + 4 P = 1), of a dense model and under
    SGP_MULTI_OWNERS=cyclic the cyclic deal (0); every panel is factored exactly once either way.  (What the table does to the
    per-rank work is checked on the host, tests/own_table_host.cpp, and measured: profiles/r05_projection_target_*.)"""
    monkeypatch.setenv("SGP_MULTI_PANEL", "128")
    F, x, xs, y = _problem(6000, D=3)
    rng = np.random.default_rng(0)
    xd = P.ColVecs(np.asfortranarray(rng.standard_normal((3, 6000))))
    fd = P.atomic(P.GP(P.SEKernel()), P.GPC())
    mode = {}
    for owners in ("cyclic", "balanced", None):
        if owners is None:
            monkeypatch.delenv("SGP_MULTI_OWNERS", raising=False)
        else:
            monkeypatch.setenv("SGP_MULTI_OWNERS", owners)
        ctx = P.lib.Context(devices=[0] * 4)
        v = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
        st = _stats(ctx, 4)
        assert sum(st[10 + 4 * i] for i in range(4)) == 47          # every panel factored exactly once
        mode[owners] = st[9 + 4 * 4]
        if owners is None:
            assert abs(v - P.logpdf(F(x, 0.1), y)) <= 1e-11 * abs(v)
            _with_ctx(ctx, lambda: P.logpdf(fd(xd, 0.1), y))
            assert _stats(ctx, 4)[9 + 4 * 4] == 0                   # one block: nothing to balance against
        ctx.close()
    assert mode["cyclic"] == 0 and mode["balanced"] == 1 and mode[None] == 1


# ---- round 6: the hybrid schedule in the sharded sweep ------------------------------------------------------------------
@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_dataflow_panel_launches_give_the_launch_based_chains_bits(monkeypatch, nranks):
    """Round 6: a sub-panel of the sharded factorisation is factored by ONE launch of the dataflow kernel that also updates the
    panel's remaining columns (update-only tile columns), and the look-ahead update with the previous panel's last sub-panel
    rides in the first of those launches as an external source (multi.hip: factor; chol_df.hip: DfExt / T_f).  Same bits as
    the launch-based chain of rounds 2 - 5 (SGP_MULTI_PANEL_DF=0) and as the unfused form (SGP_MULTI_FUSE_LA=0): logpdf with
    vector and matrix right-hand sides, kept-factor posterior, draws -- whole panels, sub-panels, panel groups, mixed widths,
    structural zeros on and off, one enqueue thread and one per rank."""
    F, x, xs, y = _problem(3100, D=3)
    ref = orm.gppp_sum_logpdf(xs, y, 0.1)
    rng = np.random.default_rng(7)
    Y = np.asfortranarray(rng.standard_normal((3100, 3)))
    xs_new = P.BlockData([P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((3, 40)))))])
    Z = np.asfortranarray(rng.standard_normal((3100, 2)))

    def run():
        fx = F(x, 0.1)
        post = P.posterior(fx, y)
        mm, vv = post.mean_and_var(xs_new)
        return dict(lp=np.array([P.logpdf(fx, y)]), lpm=np.asarray(P.logpdf(fx, Y)), m=np.asarray(mm), v=np.asarray(vv),
                    r=np.asarray(P.rand(None, fx, 2, Z=Z)))

    keys = ("SGP_MULTI_PANEL", "SGP_MULTI_SUBPANEL", "SGP_MULTI_GROUP", "SGP_MULTI_PANEL_TAIL", "SGP_STRUCT_ZEROS", "SGP_MULTI_THREADS")
    for env in ({"SGP_MULTI_PANEL": "128"},
                {"SGP_MULTI_PANEL": "512", "SGP_MULTI_SUBPANEL": "128"},
                {"SGP_MULTI_PANEL": "512", "SGP_MULTI_SUBPANEL": "256", "SGP_STRUCT_ZEROS": "0"},
                {"SGP_MULTI_PANEL": "1024", "SGP_MULTI_SUBPANEL": "0", "SGP_MULTI_THREADS": "0"},
                {"SGP_MULTI_PANEL": "256", "SGP_MULTI_SUBPANEL": "128", "SGP_MULTI_GROUP": "2"},
                {"SGP_MULTI_PANEL": "512", "SGP_MULTI_PANEL_TAIL": "256", "SGP_MULTI_SUBPANEL": "128"}):
        outs = []
        for df, fuse in (("0", "1"), ("1", "0"), ("1", "1")):
            for k in keys:
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            monkeypatch.setenv("SGP_MULTI_PANEL_DF", df)
            monkeypatch.setenv("SGP_MULTI_FUSE_LA", fuse)
            monkeypatch.setenv("SGP_MULTI_OWNERS", "cyclic")
            ctx = P.lib.Context(devices=[0] * nranks)
            outs.append(_with_ctx(ctx, run))
            outs.append(_with_ctx(ctx, run))   # (repeated call: the rings and stores are reused)
            ctx.close()
        for o in outs[1:]:
            for k in outs[0]:
                assert np.array_equal(outs[0][k], o[k]), (env, k)
        assert abs(outs[0]["lp"][0] - ref) <= 1e-10 * abs(ref)


def test_dataflow_panel_launch_reports_the_failing_minor(monkeypatch):
    """PosDef information from a panel launch of the dataflow kernel carries the GLOBAL column (gcol_base of the packed
    sub-panel), whichever rank's launch meets the bad pivot."""
    monkeypatch.setenv("SGP_MULTI_PANEL", "256")
    monkeypatch.setenv("SGP_MULTI_SUBPANEL", "128")
    F, x, xs, y = _problem(1500)
    infos = []
    for df in ("0", "1"):
        monkeypatch.setenv("SGP_MULTI_PANEL_DF", df)
        ctx = P.lib.Context(devices=[0, 0, 0])
        with pytest.raises(P.PosDefException) as e:
            _with_ctx(ctx, lambda: P.logpdf(F(x, -5.0), y))
        infos.append(e.value.info)
        ctx.close()
    assert infos[0] == infos[1] and infos[0] >= 1


@pytest.mark.parametrize("nranks", [2, 5])
def test_compacted_live_tile_ids_in_the_far_updates_keep_the_bits(monkeypatch, nranks):
    """Round 6: the far update launches of a structured model compact their live tile ids per XCD first (gemm_nt.hip:
    seg_compact_kernel -- the single-GPU launches' compaction for a list of destination panels).  Same bits with the map on
    (forced at every size), off, and against the dense schedule; the work counter is unchanged."""
    F, x, xs, y = _problem(4200, D=3)
    ref = orm.gppp_sum_logpdf(xs, y, 0.1)
    vals = []
    for env in ({"SGP_MULTI_COMPACT": "2"}, {"SGP_MULTI_COMPACT": "0"}, {"SGP_MULTI_COMPACT": "2", "SGP_STRUCT_ZEROS": "0"},
                {"SGP_MULTI_COMPACT": "2", "SGP_MULTI_GROUP": "2", "SGP_MULTI_PANEL": "256"}):
        for k in ("SGP_MULTI_COMPACT", "SGP_STRUCT_ZEROS", "SGP_MULTI_GROUP", "SGP_MULTI_PANEL"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv("SGP_MULTI_PANEL", "128")
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = P.lib.Context(devices=[0] * nranks)
        vals.append(_with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y)))
        vals.append(_with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y)))
        ctx.close()
    assert all(v == vals[0] for v in vals[:6]), vals          # one panel layout: bit for bit
    assert abs(vals[6] - vals[0]) <= 1e-12 * abs(vals[0])      # another layout: to rounding
    assert abs(vals[0] - ref) <= 1e-10 * abs(ref)


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_input_point_and_function_scale_gradients_sharded_over_the_ranks(panel128, nranks):
    """Round 6: sgp_logpdf_grad_x / _xs on a multi-GPU context (they used to run on devices[0]) -- every rank contracts its
    column slabs of G with the kernel derivatives row-side on a column window of each block pair, the per-rank sums are added
    in rank order.  Against the single-GPU results (1e-10 of the largest entry): the three-block sum model (panel boundaries
    inside blocks, a block boundary inside a panel) and a programme with two function-valued scales, one nested under the
    other and shared by two blocks (product.jl:25-48)."""
    F, x, xs, y = _problem(1411, D=3)
    ctx = P.lib.Context(devices=[0] * nranks)
    for noise in (0.1, 0.05 + np.random.default_rng(3).random(len(y))):
        g0 = P.logpdf_and_gradient(F(x, noise), y, inputs=True)
        g1 = _with_ctx(ctx, lambda: P.logpdf_and_gradient(F(x, noise), y, inputs=True))
        assert abs(g1["logpdf"] - g0["logpdf"]) <= 1e-11 * abs(g0["logpdf"])
        assert len(g0["inputs"]) == len(g1["inputs"]) >= 3
        for a0, a1 in zip(g0["inputs"], g1["inputs"]):
            assert a0.shape == a1.shape and np.abs(a1 - a0).max() <= 1e-10 * max(1.0, np.abs(a0).max())
    rng = np.random.default_rng(33)
    x1, x2 = rng.standard_normal(420), rng.standard_normal(275)
    yy = rng.standard_normal(695)
    gpc = P.GPC()
    f1 = P.atomic(P.GP(P.Matern32Kernel()), gpc)
    f2 = P.atomic(P.GP(P.SEKernel()), gpc)
    g1_ = (lambda t: 1.0 + 0.4 * float(np.sum(np.sin(t)))) * f1
    G = P.GPPP({"f1": f1, "g1": g1_, "h": (lambda t: float(np.exp(0.15 * np.sum(t)))) * (g1_ + f2)}, gpc)
    fx = G(P.BlockData([P.GPPPInput("h", x1), P.GPPPInput("g1", x2)]), 0.3)
    for kw in (dict(scales=True), dict(scales=True, inputs=True)):
        r0 = P.logpdf_and_gradient(fx, yy, **kw)
        r1 = _with_ctx(ctx, lambda: P.logpdf_and_gradient(fx, yy, **kw))
        assert abs(r1["logpdf"] - r0["logpdf"]) <= 1e-11 * abs(r0["logpdf"])
        assert len(r0["scales"]) == len(r1["scales"]) == 3
        for s0, s1 in zip(r0["scales"], r1["scales"]):
            assert np.abs(s1["d_values"] - s0["d_values"]).max() <= 1e-10 * max(1.0, np.abs(s0["d_values"]).max())
        if "inputs" in kw:
            for a0, a1 in zip(r0["inputs"], r1["inputs"]):
                assert np.abs(a1 - a0).max() <= 1e-10 * max(1.0, np.abs(a0).max())
    ctx.close()


@pytest.mark.parametrize("nranks", [2, 3, 5, 8])
def test_covariance_entry_points_sharded_over_the_ranks(nranks):
    """Round 6: sgp_kernelmatrix / sgp_kernelmatrix_diag on a multi-GPU context (they used to run on devices[0]): column chunks
    / point slices per rank, no communication.  cov(fx) keeps the symmetric spec's guarantee -- EXACTLY symmetric -- and its
    bits (the lower triangle comes from the same tile program, the upper one by transposition); cov(fx, gx) and var(fx)
    likewise; sizes that leave ranks without a chunk, block boundaries inside tiles and inside chunks."""
    ctx = P.lib.Context(devices=[0] * nranks)
    for N in (97, 700, 1411):
        F, x, xs, y = _problem(N, D=3)
        fx = F(x, 0.1)
        K0, v0 = P.cov(fx), P.var(fx)
        K1, v1 = _with_ctx(ctx, lambda: (P.cov(fx), P.var(fx)))
        assert K1.shape == (N, N) and np.array_equal(K1, K1.T)
        assert np.array_equal(K1, K0) and np.array_equal(v1, v0)
        assert np.array_equal(np.diag(K1), v1)
        # a rectangular cross-covariance between two different collections of the programme
        rng = np.random.default_rng(N)
        x2 = P.BlockData([P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((3, 333))))),
                          P.GPPPInput("f1", P.ColVecs(np.asfortranarray(rng.standard_normal((3, 130)))))])
        C0 = P.cov(fx, F(x2, 0.1))
        C1 = _with_ctx(ctx, lambda: P.cov(fx, F(x2, 0.1)))
        assert C1.shape == (N, 463) and np.array_equal(C1, C0)
    ctx.close()


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_gradient_with_dense_observation_noise_sharded_over_the_ranks(panel128, nranks):
    """Round 6: sgp_logpdf_grad with f(x, S::Matrix) on a multi-GPU context (it used to run on devices[0]): the sharded
    factorisation takes the dense Sigma_y at assembly (round 4), and the gradient with respect to it -- the cotangent
    G = (alpha alpha' - C^-1) / 2 itself, N x N -- comes back column slab by column slab from the ranks that own the panels.
    Against the single-GPU gradient: G, the term gradients, y and the mean."""
    N = 901
    F, x, xs, y = _problem(N, D=3)
    rng = np.random.default_rng(5)
    B = rng.standard_normal((N, 6))
    S = 0.05 * np.eye(N) + 0.01 * B @ B.T
    g0 = P.logpdf_and_gradient(F(x, S), y)
    ctx = P.lib.Context(devices=[0] * nranks)
    g1 = _with_ctx(ctx, lambda: P.logpdf_and_gradient(F(x, S), y))
    ctx.close()
    assert abs(g1["logpdf"] - g0["logpdf"]) <= 1e-11 * abs(g0["logpdf"])
    G0, G1 = np.asarray(g0["noise"]), np.asarray(g1["noise"])
    assert G0.shape == G1.shape == (N, N) and np.abs(G1 - G0).max() <= 1e-9 * np.abs(G0).max()
    assert np.abs(g1["y"] - g0["y"]).max() <= 1e-9 * np.abs(g0["y"]).max()
    for t0, t1 in zip(g0["terms"], g1["terms"]):
        assert abs(t1["d_coef"] - t0["d_coef"]) <= 1e-8 * max(1.0, abs(t0["d_coef"]))
        assert abs(t1["d_inscale"] - t0["d_inscale"]) <= 1e-8 * max(1.0, abs(t0["d_inscale"]))


def _close(a, b, tol, what):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.size:
        assert np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max()), (what, np.abs(a - b).max(), np.abs(a).max())


def _same_elbo_gradient(g0, g1, tol=1e-9):
    assert abs(g1["elbo"] - g0["elbo"]) <= 1e-10 * abs(g0["elbo"]), (g0["elbo"], g1["elbo"])
    for k in ("y", "mean", "noise", "z_noise", "var"):
        _close(g0[k], g1[k], tol, k)
    for k in ("zz", "xz", "xx"):
        for a0, a1 in zip(g0["_raw"][k], g1["_raw"][k]):
            _close(a0, a1, tol, "terms " + k)
    for k in ("x", "z", "zz_inputs", "xz_inputs"):
        if g0[k] is None:
            assert g1[k] is None
            continue
        assert len(g0[k]) == len(g1[k])
        for a0, a1 in zip(g0[k], g1[k]):
            _close(a0, a1, tol, k)
    if g0["scales"] is None:
        assert g1["scales"] is None
    else:
        assert len(g0["scales"]) == len(g1["scales"])
        for s0, s1 in zip(g0["scales"], g1["scales"]):
            _close(s0["d_values"], s1["d_values"], tol, "scales")


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_elbo_gradient_data_sharded_over_the_ranks(nranks):
    """Round 6: sgp_elbo_grad / _x / _xs on a multi-GPU context (they used to run on devices[0]) -- the data points are
    sharded as for the ELBO itself, every rank runs the pipeline on its slice, the sums over data points (A A', A delta, four
    scalars) meet in one reduction between the two factorisations, the M x M stage runs replicated.  Against the single-GPU
    results: a single process with isotropic and diagonal noise (every result the entry points return); a programme whose
    data blocks are split by the slices, with inducing points on two processes, input-point gradients (x points come back
    slice by slice, z points summed over the ranks); function-valued scales at x (sliced) and at z (summed)."""
    ctx = P.lib.Context(devices=[0] * nranks)
    rng = np.random.default_rng(41)
    D, N, M = 3, 1700, 130
    f = P.stretch(P.atomic(P.GP(0.3, P.Matern52Kernel()), P.GPC()), 0.7)
    X = np.asfortranarray(rng.standard_normal((D, N)))
    Z = np.asfortranarray(X[:, :M] + 0.01)
    y = rng.standard_normal(N)
    for nz in (0.1, 0.05 + rng.random(N)):
        for kw in (dict(), dict(inputs=True)):
            vfe, fx = P.VFE(f(P.ColVecs(Z), 1e-4)), f(P.ColVecs(X), nz)
            g0 = P.elbo_and_gradient(vfe, fx, y, **kw)
            g1 = _with_ctx(ctx, lambda: P.elbo_and_gradient(vfe, fx, y, **kw))
            g2 = _with_ctx(ctx, lambda: P.elbo_and_gradient(vfe, fx, y, **kw))
            _same_elbo_gradient(g0, g1)
            _same_elbo_gradient(g1, g2, tol=0.0)                 # deterministic: fixed-order reductions
            # ... and really sharded: A A' summed slice by slice rounds differently from the one-GPU split-K sum
            assert not np.array_equal(g0["y"], g1["y"])
    # a programme: three data blocks of different lengths (slice boundaries fall inside them), inducing points on f1 and f3
    F = P.gppp_sum_model()
    mats = [np.asfortranarray(rng.standard_normal((2, n))) for n in (300, 420, 515)]
    zm = [np.asfortranarray(rng.standard_normal((2, 40))), np.asfortranarray(rng.standard_normal((2, 33)))]
    xb = P.BlockData([P.GPPPInput(k, P.ColVecs(a)) for k, a in zip(("f1", "f2", "f3"), mats)])
    zb = P.BlockData([P.GPPPInput("f1", P.ColVecs(zm[0])), P.GPPPInput("f3", P.ColVecs(zm[1]))])
    yy = rng.standard_normal(len(xb))
    g0 = P.elbo_and_gradient(P.VFE(F(zb, 1e-4)), F(xb, 0.2), yy, inputs=True)
    g1 = _with_ctx(ctx, lambda: P.elbo_and_gradient(P.VFE(F(zb, 1e-4)), F(xb, 0.2), yy, inputs=True))
    _same_elbo_gradient(g0, g1)
    # function-valued scales, one nested under the other (product.jl:25-48), with dense Sigma_z
    x, z1, z2 = rng.standard_normal(900), rng.standard_normal(30), rng.standard_normal(25)
    gpc = P.GPC()
    f1 = P.atomic(P.GP(P.Matern32Kernel()), gpc)
    f2 = P.atomic(P.GP(P.SEKernel()), gpc)
    g1_ = (lambda t: 1.0 + 0.4 * float(np.sum(np.sin(t)))) * f1
    G = P.GPPP({"f1": f1, "f2": f2, "g1": g1_, "h": (lambda t: float(np.exp(0.15 * np.sum(t)))) * (g1_ + f2)}, gpc)
    B = rng.standard_normal((55, 4))
    Sz = 1e-3 * np.eye(55) + 1e-4 * B @ B.T
    vfe = P.VFE(G(P.BlockData([P.GPPPInput("g1", z1), P.GPPPInput("f2", z2)]), Sz))
    fx = G(P.GPPPInput("h", x), 0.3)
    ys = rng.standard_normal(900)
    for kw in (dict(scales=True), dict(scales=True, inputs=True)):
        r0 = P.elbo_and_gradient(vfe, fx, ys, **kw)
        r1 = _with_ctx(ctx, lambda: P.elbo_and_gradient(vfe, fx, ys, **kw))
        _same_elbo_gradient(r0, r1)
    # fewer data points than 128 per rank: not sharded, still exact
    fxs = f(P.ColVecs(np.asfortranarray(X[:, :100])), 0.1)
    g0 = P.elbo_and_gradient(P.VFE(f(P.ColVecs(Z[:, :8]), 1e-4)), fxs, y[:100])
    g1 = _with_ctx(ctx, lambda: P.elbo_and_gradient(P.VFE(f(P.ColVecs(Z[:, :8]), 1e-4)), fxs, y[:100]))
    _same_elbo_gradient(g0, g1)
    ctx.close()


def test_elbo_gradient_failure_on_the_multi_gpu_context_is_reported_and_the_context_stays_usable():
    """K(z,z) + Sigma_z not positive definite: every rank's first factorisation fails before the reduction -- nobody waits for
    anybody, the leading minor is reported, and the next call on the same context works."""
    ctx = P.lib.Context(devices=[0, 0, 0])
    rng = np.random.default_rng(42)
    N, M = 700, 40
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    X = np.asfortranarray(rng.standard_normal((2, N)))
    Zd = np.asfortranarray(np.repeat(X[:, :M // 2], 2, axis=1))          # duplicated inducing points, no jitter
    y = rng.standard_normal(N)
    with pytest.raises(P.PosDefException):
        _with_ctx(ctx, lambda: P.elbo_and_gradient(P.VFE(f(P.ColVecs(Zd), -1e-3)), f(P.ColVecs(X), 0.1), y))
    Zg = np.asfortranarray(X[:, :M] + 0.01)
    g0 = P.elbo_and_gradient(P.VFE(f(P.ColVecs(Zg), 1e-4)), f(P.ColVecs(X), 0.1), y)
    g1 = _with_ctx(ctx, lambda: P.elbo_and_gradient(P.VFE(f(P.ColVecs(Zg), 1e-4)), f(P.ColVecs(X), 0.1), y))
    _same_elbo_gradient(g0, g1)
    ctx.close()
