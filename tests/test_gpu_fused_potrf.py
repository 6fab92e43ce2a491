"""GPU tests of the fused trailing update + next-diagonal-block factorisation (SGP_FUSE_POTRF,
stheno.jl_amd/csrc/gemm_nt.hip: gemm_nt_dma_potrf_kernel; capi.hip: panel_factor / chol_bordered).

The fused launches do the SAME arithmetic as the separate update and potrf_diag launches -- the tile update, then
potrf_diag_body on the stored tile -- only in one workgroup instead of two kernels, so every operator built on the
blocked Cholesky (logpdf, posterior, rand, elbo, the gradients; LinearAlgebra.cholesky under AbstractGPs [EXT],
SURVEY 8a A2-A5) must give bit-identical results with the knob at 0, 1 (inner K = 128 updates), 2 (outer
trailing updates), 3, and with bit 3 (+8: the updated tile goes from the accumulators straight into the
factorisation's LDS layout instead of through global memory).  The oracle comparison is there so that
"identical" cannot mean "identically wrong"."""
import numpy as np
import pytest

import stheno_jl_amd as P
from oracle import reference_model as orm

pytestmark = pytest.mark.gpu


def _problem(N, D=3, seed=4242):
    rng = np.random.default_rng(seed)
    F = P.gppp_sum_model()
    n1 = N // 3
    xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (n1, n1, N - 2 * n1)]
    x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
    y = rng.standard_normal(N)
    return F, x, xs, y


def _ctx(monkeypatch, fuse, **env):
    # (these tests are about the launch-based schedules: the dataflow kernel and the hybrid schedule, the defaults from 3072
    # and 24576 columns on, have their own -- tests/test_gpu_dataflow.py)
    monkeypatch.setenv("SGP_DATAFLOW", "0")
    monkeypatch.setenv("SGP_FUSE_POTRF", str(fuse))
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    return P.lib.Context(0)


def _with_ctx(ctx, fn):
    prev = P.lib.set_default_context(ctx)
    try:
        return fn()
    finally:
        P.lib.set_default_context(prev)


def _operators(N):
    """Every operator that runs the blocked Cholesky, on one seeded problem; returns a flat dict of arrays."""
    F, x, xs, y = _problem(N)
    rng = np.random.default_rng(N)
    out = {}
    fx = F(x, 0.1)
    out["logpdf"] = np.array([P.logpdf(fx, y)])
    Y = np.asfortranarray(rng.standard_normal((N, 3)))
    out["logpdf_cols"] = np.asarray(P.logpdf(fx, Y))
    post = P.posterior(fx, y)
    out["alpha"] = np.asarray(post.alpha)
    xs_new = P.BlockData([P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((3, 40)))))])
    m, v = post.mean_and_var(xs_new)
    out["post_mean"], out["post_var"] = np.asarray(m), np.asarray(v)
    out["post_cov"] = np.asarray(post.cov(xs_new))
    Z = np.asfortranarray(rng.standard_normal((N, 2)))
    out["rand"] = np.asarray(P.rand(None, fx, 2, Z=Z))
    g = P.logpdf_and_gradient(fx, y)
    out["grad_value"] = np.array([g["logpdf"]])
    out["grad_y"] = np.asarray(g["y"])
    out["grad_noise"] = np.atleast_1d(np.asarray(g["noise"], dtype=float))
    out["grad_terms"] = np.array([[t["d_coef"], t["d_inscale"]] for t in g["terms"]])
    M = max(16, N // 7)
    z = P.BlockData([P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((3, M)))))])
    out["elbo"] = np.array([P.elbo(P.VFE(F(z, 1e-6)), fx, y)])
    return out, (xs, y)


@pytest.mark.parametrize("N", [130, 700, 4500])
def test_fused_potrf_is_bit_identical_to_separate_launches(monkeypatch, N):
    # default panel widths: N = 130 / 700 are one panel (inner fusion only), 4500 has five outer panels of 1024
    ref_ctx = _ctx(monkeypatch, 0)
    ref, (xs, y) = _with_ctx(ref_ctx, lambda: _operators(N))
    want = orm.gppp_sum_logpdf(xs, y, 0.1)
    assert abs(ref["logpdf"][0] - want) <= 1e-10 * abs(want)
    for fuse in (1, 2, 3, 9, 11):
        ctx = _ctx(monkeypatch, fuse)
        got, _ = _with_ctx(ctx, lambda: _operators(N))
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (N, fuse, k, np.max(np.abs(ref[k] - got[k])))
        ctx.close()
    ref_ctx.close()


@pytest.mark.parametrize("env", [dict(SGP_WOUT=256), dict(SGP_WOUT=128), dict(SGP_LOOKAHEAD=0, SGP_WOUT=384)])
def test_fused_potrf_with_narrow_outer_panels(monkeypatch, env):
    # many outer panels (every look-ahead column update and every serial trailing update carries a fused block),
    # W = 128: no inner update at all, every diagonal block but the first comes out of an outer fused launch
    N = 1700
    ref_ctx = _ctx(monkeypatch, 0, **env)
    ref, (xs, y) = _with_ctx(ref_ctx, lambda: _operators(N))
    want = orm.gppp_sum_logpdf(xs, y, 0.1)
    assert abs(ref["logpdf"][0] - want) <= 1e-10 * abs(want)
    for fuse in (3, 11):
        ctx = _ctx(monkeypatch, fuse, **env)
        got, _ = _with_ctx(ctx, lambda: _operators(N))
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (env, fuse, k, np.max(np.abs(ref[k] - got[k])))
        ctx.close()
    ref_ctx.close()


def test_fused_potrf_reports_the_failing_minor(monkeypatch):
    # PosDefException(info) must name the same leading minor whether the bad pivot sits in a block factored by
    # the stand-alone kernel (block 0) or by a fused launch (any later block)
    rng = np.random.default_rng(7)
    N = 600
    x = rng.standard_normal(N)
    noise = np.full(N, 0.1)
    noise[300:] = -2.0        # diagonal entry 1 - 2 < 0 at index 300: leading minor 301 fails, whatever came before
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    y = rng.standard_normal(N)
    for fuse in (0, 1, 3, 11):
        ctx = _ctx(monkeypatch, fuse)
        with pytest.raises(P.PosDefException) as e:
            _with_ctx(ctx, lambda: P.logpdf(f(x, noise), y))
        assert e.value.info == 301, (fuse, e.value.info)
        ctx.close()


@pytest.mark.parametrize("env", [dict(SGP_LOOKAHEAD=0, SGP_WOUT=1024, SGP_WMID=256), dict(SGP_LOOKAHEAD=0, SGP_WOUT=768, SGP_WMID=128),
                                 dict(SGP_WOUT=512, SGP_WMID=256)])
def test_middle_blocking_level_is_bit_identical(monkeypatch, env):
    """Round 3: outer panels factored by recursive halving down to SGP_WMID columns (capi.hip: panel_factor_mid; the
    default from 65536 columns on is 4096 / 1024).  A tile's updates are accumulated in the same ascending order of the
    contraction index whatever the grouping into launches, so every operator is bit-identical to the plain two-level
    driver of the same outer width, with the fusion on or off -- here at sizes with several outer panels, uneven halves
    (768 = 384 + 384 -> 384 + 384 rounded to multiples of the mid width) and a ragged last panel."""
    N = 1900
    base = {k: v for k, v in env.items() if k != "SGP_WMID"}
    ref_ctx = _ctx(monkeypatch, 0, **base)
    monkeypatch.delenv("SGP_WMID", raising=False)
    ref, (xs, y) = _with_ctx(ref_ctx, lambda: _operators(N))
    want = orm.gppp_sum_logpdf(xs, y, 0.1)
    assert abs(ref["logpdf"][0] - want) <= 1e-10 * abs(want)
    for fuse in (0, 11):
        ctx = _ctx(monkeypatch, fuse, **env)
        got, _ = _with_ctx(ctx, lambda: _operators(N))
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (env, fuse, k, np.max(np.abs(ref[k] - got[k])))
        ctx.close()
    ref_ctx.close()
