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
from test_gpu_fused_potrf import _ctx, _operators, _with_ctx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [130, 700, 4500])
def test_dataflow_is_bit_identical_to_the_launch_based_factorisation(monkeypatch, N):
    ref_ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0)
    ref, (xs, y) = _with_ctx(ref_ctx, lambda: _operators(N))
    want = orm.gppp_sum_logpdf(xs, y, 0.1)
    assert abs(ref["logpdf"][0] - want) <= 1e-10 * abs(want)
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=1)
    for rep in range(2):   # second pass: every cache warm with the first pass's data at the same addresses
        got, _ = _with_ctx(ctx, lambda: _operators(N))
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (N, rep, k, np.max(np.abs(ref[k] - got[k])))
    ctx.close()
    ref_ctx.close()


@pytest.mark.parametrize("wgs", [1, 3, 64, 2048])
def test_any_number_of_persistent_workgroups(monkeypatch, wgs):
    # 1 workgroup = the tasks strictly in their topological order; 2048 = far more than fit the chip (the surplus finds
    # the task counter exhausted): progress must not depend on residency
    N = 1500
    ref_ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0)
    ref, _ = _with_ctx(ref_ctx, lambda: _operators(N))
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=1, SGP_DF_WGS=wgs)
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
