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
