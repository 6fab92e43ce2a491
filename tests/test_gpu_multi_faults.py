"""Failure path of the in-process multi-GPU context (csrc/multi.hip; SURVEY.md 8b "Errors" / "Threading").  No multi-GPU
node was available in any round, so the first run on one must be unable to hang or lie: a TEST-ONLY fault hook
(sgp_bench_multi_fault / SGP_MULTI_FAULT=rank:step) makes one rank's enqueue thread fail in the middle of the schedule --
as a failing HIP or RCCL call on that thread would -- and these tests check, with several loopback ranks on the one GPU and
with the RCCL transport on one rank, that
  (a) the failing call comes back quickly (every other thread notices the abort flag in its event spin),
  (b) peer-copy / loopback contexts drain and stay usable, an RCCL context whose communicators had to be aborted reports
      `broken` and refuses further sharded calls with a clean error,
  (c) a fresh context in the same process gives the bits of the undisturbed run,
  (d) a cross-thread spin that never sees its record ends at its wall-clock bound instead of spinning for ever."""
import ctypes as C
import time

import numpy as np
import pytest

import stheno_jl_amd as P

pytestmark = pytest.mark.gpu


def _problem(N, D=3, seed=99):
    rng = np.random.default_rng(seed)
    F = P.gppp_sum_model()
    n1 = N // 3
    xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (n1, n1, N - 2 * n1)]
    x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
    return F, x, rng.standard_normal(N)


def _with_ctx(ctx, fn):
    prev = P.lib.set_default_context(ctx)
    try:
        return fn()
    finally:
        P.lib.set_default_context(prev)


def _broken(ctx):
    b = C.c_int()
    P.lib.check(ctx.bench.sgp_bench_multi_broken(ctx.handle, C.byref(b)))
    return b.value


@pytest.mark.parametrize("nranks,threads", [(2, "1"), (5, "1"), (8, "1"), (3, "0")])
@pytest.mark.parametrize("bcast", ["allgather", "direct"])
def test_injected_fault_returns_quickly_and_the_loopback_context_stays_usable(monkeypatch, nranks, threads, bcast):
    monkeypatch.setenv("SGP_MULTI_PANEL", "128")
    monkeypatch.setenv("SGP_MULTI_THREADS", threads)
    monkeypatch.setenv("SGP_MULTI_BCAST", bcast)
    F, x, y = _problem(2600)
    ctx = P.lib.Context(devices=[0] * nranks)
    good = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    for rank, step in ((nranks - 1, 7), (0, 0), (1 % nranks, 19)):
        P.lib.check(ctx.bench.sgp_bench_multi_fault(ctx.handle, rank, step))
        t0 = time.perf_counter()
        with pytest.raises(P.SthenoMIError) as e:
            _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
        assert time.perf_counter() - t0 < 10.0
        assert "injected fault" in str(e.value), str(e.value)      # the root cause, not "another rank's thread failed"
        assert not isinstance(e.value, P.PosDefException)
        assert _broken(ctx) == 0
        # the hook disarmed itself; the context drained and gives the undisturbed bits again -- logpdf and a kept factor
        assert _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y)) == good
    post = _with_ctx(ctx, lambda: P.posterior(F(x, 0.1), y))
    P.lib.check(ctx.bench.sgp_bench_multi_fault(ctx.handle, 0, 3))
    with pytest.raises(P.SthenoMIError):
        _with_ctx(ctx, lambda: P.posterior(F(x, 0.1), y))
    xs_new = P.BlockData([P.GPPPInput("f3", P.ColVecs(np.asfortranarray(np.random.default_rng(2).standard_normal((3, 17)))))])
    m1 = _with_ctx(ctx, lambda: post.mean_and_var(xs_new))    # the factor kept BEFORE the failed call is intact
    m0 = P.posterior(F(x, 0.1), y).mean_and_var(xs_new)
    np.testing.assert_allclose(m1[0], m0[0], rtol=1e-9, atol=1e-11)
    ctx.close()
    fresh = P.lib.Context(devices=[0] * nranks)
    assert _with_ctx(fresh, lambda: P.logpdf(F(x, 0.1), y)) == good
    fresh.close()


def test_rccl_context_whose_communicators_were_aborted_refuses_further_calls(monkeypatch):
    """One rank over RCCL with the per-rank enqueue thread forced on (SGP_MULTI_THREADS=1): a failing thread means that some
    ranks' broadcasts may be enqueued without their partners', so the communicators are aborted and the context is marked."""
    monkeypatch.setenv("SGP_MULTI_PANEL", "128")
    monkeypatch.setenv("SGP_MULTI_TRANSPORT", "rccl")
    monkeypatch.setenv("SGP_MULTI_THREADS", "1")
    F, x, y = _problem(1500)
    ctx = P.lib.Context(devices=[0])
    assert ctx.transport == "rccl"
    good = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    assert abs(good - P.logpdf(F(x, 0.1), y)) <= 1e-11 * abs(good)
    P.lib.check(ctx.bench.sgp_bench_multi_fault(ctx.handle, 0, 5))
    t0 = time.perf_counter()
    with pytest.raises(P.SthenoMIError) as e:
        _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    assert time.perf_counter() - t0 < 10.0
    assert "injected fault" in str(e.value) and "aborted" in str(e.value)
    assert _broken(ctx) == 1
    for _ in range(2):
        with pytest.raises(P.SthenoMIError) as e2:
            _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
        assert "create a new context" in str(e2.value)
    ctx.close()
    fresh = P.lib.Context(devices=[0])
    assert _with_ctx(fresh, lambda: P.logpdf(F(x, 0.1), y)) == good
    fresh.close()
    # the one-thread enqueue issues a grouped broadcast for all ranks or for none: the communicators survive a fault
    monkeypatch.setenv("SGP_MULTI_THREADS", "0")
    ctx = P.lib.Context(devices=[0])
    P.lib.check(ctx.bench.sgp_bench_multi_fault(ctx.handle, 0, 2))
    with pytest.raises(P.SthenoMIError):
        _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    assert _broken(ctx) == 0
    assert _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y)) == good
    ctx.close()


def test_fault_from_the_environment_and_the_spin_bound(monkeypatch):
    """SGP_MULTI_FAULT=rank:step arms the hook at context creation (how a command line is tested).  And the wall-clock bound of
    the cross-thread spins: rank 1's enqueue thread SLEEPS at panel 3 (sgp_bench_multi_stall) -- it has not failed, the abort
    flag stays down, the other threads sit in Exec::wait for records it has not issued.  With SGP_MULTI_SPIN_TIMEOUT_S = 0.5
    they give up after half a second, raise the flag, and the call fails with a text that names the wait -- instead of
    spinning for as long as the stall lasts.  A normal run under the same small bound is undisturbed."""
    monkeypatch.setenv("SGP_MULTI_PANEL", "128")
    monkeypatch.setenv("SGP_MULTI_FAULT", "1:4")
    F, x, y = _problem(1800)
    ctx = P.lib.Context(devices=[0, 0, 0])
    with pytest.raises(P.SthenoMIError) as e:
        _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    assert "injected fault at panel 4" in str(e.value)
    v = _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))      # fired once
    ctx.close()
    monkeypatch.delenv("SGP_MULTI_FAULT")
    monkeypatch.setenv("SGP_MULTI_SPIN_TIMEOUT_S", "0.5")
    ctx = P.lib.Context(devices=[0, 0, 0])
    assert _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y)) == v
    P.lib.check(ctx.bench.sgp_bench_multi_stall(ctx.handle, 1, 3, 3.0))
    t0 = time.perf_counter()
    with pytest.raises(P.SthenoMIError) as e:
        _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y))
    dt = time.perf_counter() - t0
    assert "waited more than" in str(e.value) and "SGP_MULTI_SPIN_TIMEOUT_S" in str(e.value), str(e.value)
    assert dt < 10.0          # (the stalled thread itself wakes after 3 s, sees the flag and leaves)
    assert _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), y)) == v
    ctx.close()
