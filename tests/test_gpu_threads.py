"""The threading clause of the C-ABI ("a ctx serialises its own calls; distinct ctxs may be used concurrently;
no global state except thread-local error text"), exercised from several host threads -- ctypes releases the GIL around a
foreign call, so Python threads really are concurrent callers, as `Threads.@threads` over restarts is from Julia.

  (a) four threads on the SAME context, each calling sgp_logpdf at sizes that take the launch-based schedule, the one-workgroup-
      per-CU dataflow kernel and the dataflow kernel (N = 700 / 3300 / 5200 ... whatever sgp_ctx_factor_schedule says), in
      different orders: every result bit-equal to the serial run; a failing call's text arrives in ITS thread's
      sgp_last_error while the others keep succeeding;
  (b) two contexts on the same device from two threads at once, one of them forced onto the dataflow schedule, structural
      zeros on (the 3-block sum model): bit-equal to the serial run, and no caller ever sees the dataflow kernel's wait
      bound (a timeout is rerun on the launch-based schedule inside the library: capi.hip with_df_fallback -- the counter
      of such reruns is printed);
  (c) sgp_posterior_predict on ONE kept posterior from two threads."""
import threading

import numpy as np
import pytest

import stheno_jl_amd as P
from stheno_jl_amd import finite_gp as fg
from stheno_jl_amd import lib as L

pytestmark = pytest.mark.gpu


def _problem(N, D=3, seed=11):
    rng = np.random.default_rng(seed + N)
    F = P.gppp_sum_model()
    n1 = N // 3
    xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (n1, n1, N - 2 * n1)]
    x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
    y = rng.standard_normal(N)
    return F, x, y


class _Call:
    """One prepared sgp_logpdf call (spec + buffers kept alive), runnable on any context from any thread."""

    def __init__(self, N, noise=0.1):
        F, x, y = _problem(N)
        self.N = N
        self.spec, m, self.kind, self.nbuf = fg._spec_mean_noise(F(x, noise))
        self.m = np.ascontiguousarray(m, dtype=np.float64)
        self.Y = np.asfortranarray(y.reshape(N, 1))

    def run(self, ctx):
        out = np.zeros(1)
        rc = ctx.lib.sgp_logpdf(ctx.handle, self.spec.ref(), L.dptr(self.m), self.kind, L.dptr(self.nbuf), L.dptr(self.Y),
                                self.N, 1, L.dptr(out))
        return rc, out[0], (L.last_error() if rc else "")


def _run_threads(fns):
    res, errs = [None] * len(fns), []

    def wrap(i):
        try:
            res[i] = fns[i]()
        except BaseException as e:   # noqa: BLE001 - reported below
            errs.append((i, repr(e)))

    th = [threading.Thread(target=wrap, args=(i,)) for i in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not any(t.is_alive() for t in th), "a caller thread is stuck"
    assert not errs, errs
    return res


def test_four_threads_share_one_context():
    ctx = L.Context(0)
    sizes = (700, 3300, 5200)
    calls = {n: _Call(n) for n in sizes}
    bad = _Call(900, noise=-5.0)                       # not positive definite: rc > 0 and an error text
    schedules = {n: ctx.factor_schedule(n) for n in sizes}
    assert len(set(schedules.values())) >= 2, schedules   # the sizes really take different schedules
    serial = {n: calls[n].run(ctx) for n in sizes}
    assert all(rc == 0 for rc, _, _ in serial.values())
    rc_bad, _, msg_bad = bad.run(ctx)
    assert rc_bad > 0 and "positive definite" in msg_bad

    def worker(tid):
        def fn():
            got = []
            order = [sizes[(tid + j) % 3] for j in range(9)]
            for j, n in enumerate(order):
                if tid == 1 and j % 3 == 1:            # thread 1 also makes failing calls between the others' good ones
                    rc, _, msg = bad.run(ctx)
                    got.append(("bad", rc, msg))
                rc, v, msg = calls[n].run(ctx)
                got.append((n, rc, v, msg))
            return got
        return fn

    res = _run_threads([worker(t) for t in range(4)])
    for tid, got in enumerate(res):
        for rec in got:
            if rec[0] == "bad":
                assert rec[1] == rc_bad and rec[2] == msg_bad, (tid, rec)
            else:
                n, rc, v, msg = rec
                assert rc == 0 and msg == "" and v == serial[n][1], (tid, rec, serial[n])
    ctx.close()


def test_two_contexts_on_one_device_run_concurrently(monkeypatch):
    import ctypes as C
    monkeypatch.setenv("SGP_DATAFLOW", "1")            # read at sgp_ctx_create: context A factors everything by dataflow
    ctx_a = L.Context(0)
    monkeypatch.delenv("SGP_DATAFLOW")
    ctx_b = L.Context(0)
    sizes = (5200, 9000, 2500)
    calls = {n: _Call(n) for n in sizes}
    assert "dataflow" in ctx_a.factor_schedule(2500) and "dataflow" in ctx_a.factor_schedule(9000)
    serial = {n: calls[n].run(ctx_b) for n in sizes}
    assert all(rc == 0 for rc, _, _ in serial.values())
    for n in sizes:                                     # every schedule gives the same bits
        assert calls[n].run(ctx_a)[:2] == (0, serial[n][1])
    e, d = ctx_b.factor_work()
    assert e < 0.8 * d                                  # structural zeros are on (f1 and f2 are independent)

    def worker(ctx, shift):
        def fn():
            return [(n,) + calls[n].run(ctx) for j in range(8) for n in (sizes[(j + shift) % 3],)]
        return fn

    res = _run_threads([worker(ctx_a, 0), worker(ctx_b, 1), worker(ctx_a, 2)])
    for got in res:
        for n, rc, v, msg in got:
            assert rc == 0 and msg == "" and v == serial[n][1], (n, rc, v, msg, serial[n])
    fb = C.c_int64()
    L.check(ctx_a.bench.sgp_bench_df_fallbacks(ctx_a.handle, C.byref(fb)))
    print("dataflow launches rerun on the launch-based schedule while sharing the device:", fb.value)
    ctx_a.close()
    ctx_b.close()


def test_two_contexts_under_the_hybrid_schedule():
    """From 24576 columns on a factorisation launches one persistent panel kernel per 2048 columns beside its update launches
    (capi.hip: use_hybrid).  Two contexts + a second thread on one of them, all factoring at once: bit-equal to the serial run;
    the number of operators that had to be rerun on the launches after a wait bound is printed (0 on an otherwise idle box)."""
    import ctypes as C
    a, b = L.Context(0), L.Context(0)
    call = _Call(24700)
    assert a.factor_schedule(24700) == "hybrid"
    ref = call.run(a)
    assert ref[0] == 0 and call.run(b) == ref

    def worker(ctx):
        return lambda: [call.run(ctx) for _ in range(4)]

    for got in _run_threads([worker(a), worker(b), worker(a)]):
        assert all(r == ref for r in got), (got, ref)
    for ctx in (a, b):
        fb = C.c_int64()
        L.check(ctx.bench.sgp_bench_df_fallbacks(ctx.handle, C.byref(fb)))
        print("operators rerun on the launches:", fb.value)
        ctx.close()


def test_two_threads_predict_on_one_posterior():
    F, x, y = _problem(2600)
    rng = np.random.default_rng(3)
    post = P.posterior(F(x, 0.1), y)
    queries = [P.BlockData([P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((3, n)))))]) for n in (40, 300, 7)]
    serial = [tuple(np.asarray(a) for a in post.mean_and_var(q)) for q in queries]

    def worker(shift):
        def fn():
            out = []
            for j in range(6):
                k = (j + shift) % 3
                m, v = post.mean_and_var(queries[k])
                out.append((k, np.asarray(m), np.asarray(v)))
            return out
        return fn

    for got in _run_threads([worker(0), worker(1)]):
        for k, m, v in got:
            assert np.array_equal(m, serial[k][0]) and np.array_equal(v, serial[k][1]), k
