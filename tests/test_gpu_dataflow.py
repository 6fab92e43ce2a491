"""GPU tests of the dataflow (single-launch) blocked Cholesky (stheno.jl_amd/csrc/chol_df.hip, SGP_DATAFLOW).

One launch of persistent workgroups replaces the launch-per-step schedules of capi.hip: chol_bordered: tasks = 128 x 128
tiles in column-major (topological) order, tile-row progress counters instead of kernel boundaries.  Every entry of the
factor sees the same operations in the same order as on the launch-based path -- the production GEMM tile program with
k ascending, potrf_diag_body, the refined 16-row substitution -- so every operator built on the factorisation (logpdf,
posterior, rand, elbo; LinearAlgebra.cholesky under AbstractGPs [EXT]) must come out BIT-IDENTICAL.
The oracle comparison is there so that "identical" cannot mean "identically wrong"."""
import numpy as np
import pytest

import stheno_jl_amd as P
from oracle import reference_model as orm
from test_gpu_fused_potrf import _ctx, _operators, _problem, _with_ctx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fat", [0, 1])
@pytest.mark.parametrize("N", [130, 700, 4500])
def test_dataflow_is_bit_identical_to_the_launch_based_factorisation(monkeypatch, N, fat):
    # fat = 1: the one-workgroup-per-CU instantiation (256 VGPRs, chol_dataflow_fat_kernel), 0: two per CU
    ref_ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0)
    ref, (xs, y) = _with_ctx(ref_ctx, lambda: _operators(N))
    want = orm.gppp_sum_logpdf(xs, y, 0.1)
    assert abs(ref["logpdf"][0] - want) <= 1e-10 * abs(want)
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=1, SGP_DF_FAT_MAX_N=(1 << 30) if fat else 0)
    assert ctx.factor_schedule(N) == ("dataflow-fat" if fat else "dataflow")
    for rep in range(2):   # second pass: every cache warm with the first pass's data at the same addresses
        got, _ = _with_ctx(ctx, lambda: _operators(N))
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (N, rep, k, np.max(np.abs(ref[k] - got[k])))
    ctx.close()
    ref_ctx.close()


@pytest.mark.parametrize("fat", [0, 1])
@pytest.mark.parametrize("wgs", [1, 3, 64, 2048])
def test_any_number_of_persistent_workgroups(monkeypatch, wgs, fat):
    # 1 workgroup = the tasks strictly in their topological order; 2048 = far more than fit the chip (the surplus finds
    # the task counter exhausted): progress must not depend on residency
    N = 1500
    ref_ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0)
    ref, _ = _with_ctx(ref_ctx, lambda: _operators(N))
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=1, SGP_DF_WGS=wgs, SGP_DF_FAT_MAX_N=(1 << 30) if fat else 0)
    got, _ = _with_ctx(ctx, lambda: _operators(N))
    for k in ref:
        assert np.array_equal(ref[k], got[k]), (wgs, k)
    ctx.close()
    ref_ctx.close()


def test_not_positive_definite_is_reported_with_the_same_minor(monkeypatch):
    rng = np.random.default_rng(5)
    x = P.ColVecs(np.asfortranarray(rng.standard_normal((2, 600))))
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    y = rng.standard_normal(600)
    msgs = []
    for df in (0, 1):
        ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=df)
        with pytest.raises(P.lib.SthenoMIError) as e:
            _with_ctx(ctx, lambda: P.logpdf(f(x, -0.5), y))
        msgs.append(str(e.value))
        ctx.close()
    assert "not positive definite" in msgs[0] and msgs[0] == msgs[1]


def test_schedule_by_size_and_the_baseline_goldens_under_it():
    """The default context picks the schedule by the number of columns (capi.hip: use_dataflow); the committed CPU goldens
    of the BASELINE sizes that fall into the dataflow range hold under it (tests/test_gpu_baseline_golden.py runs every
    size on the default context as well -- this one pins WHICH code produced the number)."""
    import bench_configs as bc
    ctx = P.lib.default_context()
    assert ctx.factor_schedule(2048) == "launches-one-panel"
    assert ctx.factor_schedule(4096) == "dataflow-fat"
    assert ctx.factor_schedule(16384) == "dataflow-fat"
    assert ctx.factor_schedule(20000) == "dataflow-fat" or ctx.factor_schedule(20000) == "dataflow"
    assert ctx.factor_schedule(32768) == "hybrid"
    assert ctx.factor_schedule(65536) == "hybrid"
    for name in ("n4k", "c2"):
        w = bc.build(P, name)
        got = P.logpdf(w["fx"], w["y"])
        want = bc.golden(name)["logpdf"]
        assert abs(got - want) <= 1e-10 * abs(want), (name, got, want)


def test_many_right_hand_sides_and_posterior_rows_ride_through(monkeypatch):
    # bordered rows: 300 data columns (3 border tile rows) and a posterior's alpha row at N = 3500 -- tile rows below the
    # square part are tasks like any other
    rng = np.random.default_rng(11)
    N = 3500
    x = P.ColVecs(np.asfortranarray(rng.standard_normal((3, N))))
    f = P.atomic(P.GP(P.Matern32Kernel()), P.GPC())
    Y = np.asfortranarray(rng.standard_normal((N, 300)))
    outs = []
    for df in (0, 1):
        ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=df)
        outs.append(_with_ctx(ctx, lambda: (np.asarray(P.logpdf(f(x, 0.2), Y)), np.asarray(P.posterior(f(x, 0.2), Y[:, 0]).alpha))))
        ctx.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_a_dependency_wait_that_runs_out_is_an_error_not_a_hang(monkeypatch):
    # every wait inside the kernel is bounded (SGP_DF_TIMEOUT_S); with a bound of 0.1 us the first tile that has to wait for
    # the diagonal block's 30 us factorisation gives up, raises the abort word (all workgroups leave) and the entry point
    # reports it (SGP_DF_FALLBACK=0: without the rerun on the launch-based schedule, tested further down); the context stays
    # usable
    rng = np.random.default_rng(3)
    x = P.ColVecs(np.asfortranarray(rng.standard_normal((2, 1500))))
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    y = rng.standard_normal(1500)
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=1, SGP_DF_TIMEOUT_S="1e-7", SGP_DF_FALLBACK=0)
    with pytest.raises(P.lib.SthenoMIError, match="dependency wait"):
        _with_ctx(ctx, lambda: P.logpdf(f(x, 0.1), y))
    ctx.close()
    monkeypatch.delenv("SGP_DF_FALLBACK")
    ok = _ctx(monkeypatch, 11, SGP_DATAFLOW=1, SGP_DF_TIMEOUT_S="10")
    ref = _ctx(monkeypatch, 11, SGP_DATAFLOW=0)
    assert _with_ctx(ok, lambda: P.logpdf(f(x, 0.1), y)) == _with_ctx(ref, lambda: P.logpdf(f(x, 0.1), y))
    ok.close()
    ref.close()


def test_schedule_limits_move_with_their_environment_variables(monkeypatch):
    # SGP_DATAFLOW unset / -1 = by size; SGP_DF_MIN_N, SGP_DF_MAX_N, SGP_DF_FAT_MAX_N move the limits
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=-1, SGP_DF_MIN_N=1024, SGP_DF_MAX_N=8192, SGP_DF_FAT_MAX_N=4096)
    assert ctx.factor_schedule(896) == "launches-one-panel"
    assert ctx.factor_schedule(1000) == "dataflow-fat"              # the limits apply to the padded column count (1024)
    assert ctx.factor_schedule(4096) == "dataflow"
    assert ctx.factor_schedule(8192) == "launches-lookahead"
    assert ctx.factor_schedule(24576) == "hybrid"
    ctx.close()
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=-1, SGP_HYBRID_MIN_N=6100)
    assert ctx.factor_schedule(6000) == "dataflow" and ctx.factor_schedule(6100) == "hybrid"   # (6016 / 6144 padded columns)
    ctx.close()
    monkeypatch.delenv("SGP_HYBRID_MIN_N")
    off = _ctx(monkeypatch, 11, SGP_DATAFLOW=-1, SGP_HYBRID=0)
    assert off.factor_schedule(32768) == "launches-lookahead" and off.factor_schedule(65536) == "launches-serial-deep"
    off.close()
    monkeypatch.delenv("SGP_HYBRID")
    never = _ctx(monkeypatch, 11, SGP_DATAFLOW=0)
    always = _ctx(monkeypatch, 11, SGP_DATAFLOW=1)
    for N in (130, 5000, 70000):
        assert not never.factor_schedule(N).startswith("dataflow")
        assert always.factor_schedule(N).startswith("dataflow")
    never.close()
    always.close()


@pytest.mark.parametrize("fat,wgs,W", [(1, 256, 1024), (0, 512, 1024), (1, 7, 2048), (0, 96, 512)])
@pytest.mark.parametrize("N", [4500, 6200])
def test_hybrid_schedule_is_bit_identical_to_the_launch_based_factorisation(monkeypatch, N, fat, wgs, W):
    """The hybrid schedule (capi.hip: use_hybrid; the default from 24576 columns on, forced here at test sizes): every outer
    panel factored by one launch of the dataflow kernel ON THE PANEL (its pattern and PosDef column offset by the panel's
    position), the trailing updates lock-step launches with look-ahead.  The three-block sum model has structural zeros:
    the panel launches read the whole matrix's tile pattern at their offset."""
    ref_ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0, SGP_HYBRID=0)
    ref, (xs, y) = _with_ctx(ref_ctx, lambda: _operators(N))
    want = orm.gppp_sum_logpdf(xs, y, 0.1)
    assert abs(ref["logpdf"][0] - want) <= 1e-10 * abs(want)
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=-1, SGP_HYBRID=1, SGP_HYBRID_W=W, SGP_HYBRID_WGS=wgs, SGP_HYBRID_FAT=fat)
    assert ctx.factor_schedule(N) == "hybrid"
    for rep in range(2):
        got, _ = _with_ctx(ctx, lambda: _operators(N))
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (N, rep, k, np.max(np.abs(ref[k] - got[k])))
    F, x, _, yy = _problem(N)
    _with_ctx(ctx, lambda: P.logpdf(F(x, 0.1), yy))
    e, d = ctx.factor_work()
    assert e < 0.9 * d          # the structure is on under the hybrid schedule
    for sz in (0,):             # ... and the same bits without it
        dense = _ctx(monkeypatch, 11, SGP_DATAFLOW=-1, SGP_HYBRID=1, SGP_HYBRID_W=W, SGP_STRUCT_ZEROS=sz)
        got, _ = _with_ctx(dense, lambda: _operators(N))
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (N, "dense", k)
        dense.close()
    ctx.close()
    ref_ctx.close()


def test_hybrid_reports_the_same_failing_minor_and_survives_a_timeout(monkeypatch):
    import ctypes as C
    rng = np.random.default_rng(5)
    N = 5000
    x = P.ColVecs(np.asfortranarray(rng.standard_normal((2, N))))
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    y = rng.standard_normal(N)
    msgs = []
    for hy in (0, 1):   # SE kernel with negative noise: the failing minor lies in a later panel or the first, the text says which
        ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0, SGP_HYBRID=hy, SGP_HYBRID_W=1024)
        with pytest.raises(P.lib.SthenoMIError) as e:
            _with_ctx(ctx, lambda: P.logpdf(f(x, -1e-3), y))
        msgs.append(str(e.value))
        ctx.close()
    assert "not positive definite" in msgs[0] and msgs[0] == msgs[1], msgs
    # a wait bound of a nanosecond: the operator is rerun on the launch-based schedule, same bits
    ref_ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0, SGP_HYBRID=0)
    ref, _ = _with_ctx(ref_ctx, lambda: _operators(4500))
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=-1, SGP_HYBRID=1, SGP_HYBRID_W=1024, SGP_DF_TIMEOUT_S="1e-9")
    got, _ = _with_ctx(ctx, lambda: _operators(4500))
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k
    n = C.c_int64()
    P.lib.check(ctx.bench.sgp_bench_df_fallbacks(ctx.handle, C.byref(n)))
    assert n.value >= 1
    ctx.close()
    ref_ctx.close()


def test_a_dataflow_timeout_falls_back_to_the_launches_with_the_same_bits(monkeypatch):
    """Advisor, round 3: the dataflow kernel bounds its inter-workgroup waits by wall-clock time and used to fail the
    operator (rc -3) when a wait ran out -- under preemption / profiler serialisation nothing is wrong, the launch-based
    schedule would simply have run slower.  With a bound of a nanosecond every dependency wait "times out"; the entry
    points rerun the operator on the launch-based schedule (capi.hip: with_df_fallback): same bits, no error."""
    import ctypes as C
    N = 4500
    ref_ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0)
    ref, _ = _with_ctx(ref_ctx, lambda: _operators(N))
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=1, SGP_DF_TIMEOUT_S="1e-9")
    got, _ = _with_ctx(ctx, lambda: _operators(N))
    for k in ref:
        assert np.array_equal(ref[k], got[k]), k
    n = C.c_int64()
    P.lib.check(ctx.bench.sgp_bench_df_fallbacks(ctx.handle, C.byref(n)))
    assert n.value >= 1, "no wait ran into the 1 ns bound?"
    ctx.close()
    ref_ctx.close()


def test_tall_bordered_vfe_shape_is_bit_identical_under_both_schedules(monkeypatch):
    """Advisor, round 3: the VFE pipeline sends K(z,z) with ALL data rows as bordered rows through chol_bordered when it is
    not chunked -- M >= 3072 inducing points selects the dataflow kernel with hundreds of bordered tile rows (T_r >> T_c),
    a shape no other test reaches.  ELBO and the sparse posterior must come out bit-identical with SGP_DATAFLOW = 0 / 1."""
    rng = np.random.default_rng(17)
    N, M, D = 20000, 3200, 3
    X = np.asfortranarray(rng.standard_normal((D, N)))
    Z = np.asfortranarray(X[:, rng.permutation(N)[:M]] + 0.01 * rng.standard_normal((D, M)))
    y = rng.standard_normal(N)
    xs = P.ColVecs(np.asfortranarray(rng.standard_normal((D, 50))))
    f = P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 0.7)
    out = []
    for df in (0, 1):
        ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=df, SGP_VFE_CHUNK=0)
        assert ctx.factor_schedule(M) == ("launches-one-panel" if df == 0 else "dataflow-fat")

        def run():
            vfe = P.VFE(f(P.ColVecs(Z), 1e-6))
            e = P.elbo(vfe, f(P.ColVecs(X), 0.1), y)
            m, v = P.posterior(vfe, f(P.ColVecs(X), 0.1), y).mean_and_var(xs)
            return e, m, v
        out.append(_with_ctx(ctx, run))
        ctx.close()
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
