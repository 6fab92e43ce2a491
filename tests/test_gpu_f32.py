"""GPU tests of the fp32 instantiation (sgp_logpdf_f32 / sgp_kernelmatrix_f32, stheno.jl_amd/csrc/f32.hip):
the reference is type-stable in Float32 (/root/reference/test/gp/util.jl:76-88): Float32 inputs and
observations give Float32 results.  Values are compared with the fp64 oracle at fp32 tolerances."""
import numpy as np
import pytest

import models
import oracle.abstractgps as oagp
import oracle.kernelfunctions as okf
import oracle.stheno as ost
import stheno_jl_amd as P

pytestmark = pytest.mark.gpu


def _both(recipe):
    fo, go = recipe(models.oracle_api())
    fp, gp = recipe(models.product_api())
    return ost.GPPP(fo, go), P.GPPP(fp, gp)


@pytest.mark.parametrize("kind", ["se", "matern52", "matern32", "matern12"])
@pytest.mark.parametrize("N", [1, 127, 128, 129, 700, 2500])
def test_logpdf_f32_single_gp(N, kind):
    rng = np.random.default_rng(N + len(kind))
    D = 3
    ko = {"se": okf.SEKernel, "matern52": okf.Matern52Kernel, "matern32": okf.Matern32Kernel,
          "matern12": okf.Matern12Kernel}[kind]()
    kp = {"se": P.SEKernel, "matern52": P.Matern52Kernel, "matern32": P.Matern32Kernel,
          "matern12": P.Matern12Kernel}[kind]()
    X32 = rng.standard_normal((D, N)).astype(np.float32)
    y32 = rng.standard_normal(N).astype(np.float32)
    fo = ost.atomic(oagp.GP(0.3, ko), ost.GPC())
    fp = P.atomic(P.GP(0.3, kp), P.GPC())
    ref = oagp.logpdf(fo(okf.ColVecs(X32.astype(np.float64)), 0.1), y32.astype(np.float64))
    lp = P.logpdf(fp(P.ColVecs(X32), np.float32(0.1)), y32)
    assert isinstance(lp, np.float32)                       # type stability (test/gp/util.jl:76-88)
    assert abs(float(lp) - ref) <= 1e-4 * max(1.0, abs(ref)), (lp, ref)
    # the same call with Float64 data stays on the fp64 path
    lp64 = P.logpdf(fp(P.ColVecs(X32.astype(np.float64)), 0.1), y32.astype(np.float64))
    assert isinstance(lp64, float) and abs(lp64 - ref) <= 1e-10 * max(1.0, abs(ref))


def test_logpdf_f32_gppp_blocks_diag_noise_and_means():
    rng = np.random.default_rng(8)
    Fo, Fp = _both(models.toy_gppp)
    xs = [rng.standard_normal(n).astype(np.float32) for n in (300, 257, 411)]
    names = ("f1", "f2", "f3")
    xo = ost.BlockData([ost.GPPPInput(k, v.astype(np.float64)) for k, v in zip(names, xs)])
    xp = P.BlockData([P.GPPPInput(k, v) for k, v in zip(names, xs)])
    N = len(xp)
    y = rng.standard_normal(N).astype(np.float32)
    noise = (0.05 + rng.random(N)).astype(np.float32)
    ref = oagp.logpdf(Fo(xo, noise.astype(np.float64)), y.astype(np.float64))
    lp = P.logpdf(Fp(xp, noise), y)
    assert isinstance(lp, np.float32) and abs(float(lp) - ref) <= 1e-4 * abs(ref)


def test_cov_and_mean_f32():
    rng = np.random.default_rng(9)
    Fo, Fp = _both(models.composite_kernels)
    X = rng.standard_normal((2, 190)).astype(np.float32)
    xo = ost.GPPPInput("s", okf.ColVecs(X.astype(np.float64)))
    xp = P.GPPPInput("s", P.ColVecs(X))
    Kp = P.prior_cov(Fp, xp)
    assert Kp.dtype == np.float32 and np.array_equal(Kp, Kp.T)
    assert np.abs(Kp - Fo.cov(xo)).max() < 5e-6 * np.abs(Fo.cov(xo)).max()
    Z = rng.standard_normal((2, 33)).astype(np.float32)
    Kc = P.prior_cov(Fp, xp, P.GPPPInput("f1", P.ColVecs(Z)))
    assert Kc.dtype == np.float32 and Kc.shape == (190, 33)
    assert np.abs(Kc - Fo.cov(xo, ost.GPPPInput("f1", okf.ColVecs(Z.astype(np.float64))))).max() < 1e-5
    fx = Fp(xp, np.float32(0.1))
    assert P.mean(fx).dtype == np.float32 and P.cov(fx).dtype == np.float32


def test_posdef_failure_f32():
    rng = np.random.default_rng(2)
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    X = rng.standard_normal((2, 300)).astype(np.float32)
    with pytest.raises(P.PosDefException):
        P.logpdf(f(P.ColVecs(X), np.float32(-5.0)), rng.standard_normal(300).astype(np.float32))


def test_rand_f32_type_stable_and_close_to_fp64():
    """rand(rng, fx) / rand(rng, fx, S) of a Float32 model (test/gp/util.jl:76-88 checks `rand` type stability): the
    fp32 factor times the caller's draw, against the oracle's m + L Z at fp32 tolerance."""
    rng = np.random.default_rng(12)
    Fo, Fp = _both(models.toy_gppp)
    xs = [rng.standard_normal(n).astype(np.float32) for n in (150, 129, 200)]
    names = ("f1", "f2", "f3")
    xo = ost.BlockData([ost.GPPPInput(k, v.astype(np.float64)) for k, v in zip(names, xs)])
    xp = P.BlockData([P.GPPPInput(k, v) for k, v in zip(names, xs)])
    N = len(xp)
    Z = rng.standard_normal((N, 3))
    ref = oagp.rand(Fo(xo, 0.1), Z)
    got = P.rand(None, Fp(xp, np.float32(0.1)), 3, Z=Z)
    assert got.dtype == np.float32 and got.shape == (N, 3)
    assert np.max(np.abs(got - ref)) <= 2e-4 * np.max(np.abs(ref))
    one = P.rand(None, Fp(xp, np.float32(0.1)), Z=Z[:, :1])
    assert one.dtype == np.float32 and one.shape == (N,) and np.array_equal(one, got[:, 0])


def test_posterior_moments_f32_and_fp64_factor_on_demand():
    """posterior(fx, y) of a Float32 model factors ONCE, at construction (the fp64 factor; advisor, round 3), and every
    accessor -- mean, var, cov, alpha, repeated, in any order -- runs against it with Float32 results; the one-shot fp32
    path (ONE fp32 factorisation with x* riding along as bordered rows, sgp_posterior_mean_var_f32) is the explicit
    posterior_mean_and_var_f32."""
    rng = np.random.default_rng(13)
    Fo, Fp = _both(models.gppp_docstring)
    x1, x3 = rng.standard_normal(300).astype(np.float32), rng.standard_normal(417).astype(np.float32)
    xo = ost.BlockData([ost.GPPPInput("f1", x1.astype(np.float64)), ost.GPPPInput("f3", x3.astype(np.float64))])
    xp = P.BlockData([P.GPPPInput("f1", x1), P.GPPPInput("f3", x3)])
    y = rng.standard_normal(717).astype(np.float32)
    t = np.linspace(-2, 2, 70).astype(np.float32)
    s2 = float(np.float32(0.1))          # the Float32 noise value, exactly, on both sides
    po = oagp.posterior(Fo(xo, s2), y.astype(np.float64))
    pp = P.posterior(Fp(xp, np.float32(0.1)), y)
    assert pp._h is not None             # factored at construction
    for proc in ("f2", "f3"):
        mo, vo = po.mean_and_var(ost.GPPPInput(proc, t.astype(np.float64)))
        mp, vp = pp.mean_and_var(P.GPPPInput(proc, t))
        assert mp.dtype == np.float32 and vp.dtype == np.float32
        assert np.max(np.abs(mp - mo)) <= 1e-6 * max(1.0, np.max(np.abs(mo)))     # fp64 arithmetic, rounded once
        assert np.max(np.abs(vp - vo)) <= 1e-6
        # the explicit one-shot fp32 path: fp32 accuracy
        m1, v1 = P.posterior_mean_and_var_f32(Fp(xp, np.float32(0.1)), y, P.GPPPInput(proc, t))
        assert m1.dtype == np.float32 and v1.dtype == np.float32
        assert np.max(np.abs(m1 - mo)) <= 2e-4 * max(1.0, np.max(np.abs(mo)))
        assert np.max(np.abs(v1 - vo)) <= 2e-4
    before = pp.mean_and_var(P.GPPPInput("f3", t))
    m, v = P.mean_and_var(pp(P.GPPPInput("f3", t), np.float32(0.01)))        # FiniteGP on top of the posterior
    assert m.dtype == np.float32 and v.dtype == np.float32
    c = pp.cov(P.GPPPInput("f3", t))                # the same factor; Float32 result
    assert c.dtype == np.float32
    assert np.max(np.abs(c - po.cov(ost.GPPPInput("f3", t.astype(np.float64))))) <= 1e-6
    c64 = pp.cov(P.GPPPInput("f3", t.astype(np.float64)))                     # asked at Float64 points: Float64 result
    assert c64.dtype == np.float64 and np.max(np.abs(c64 - po.cov(ost.GPPPInput("f3", t.astype(np.float64))))) <= 1e-8
    assert np.max(np.abs(pp.alpha - po.alpha)) <= 1e-8 * np.max(np.abs(po.alpha))
    after = pp.mean_and_var(P.GPPPInput("f3", t))   # a result does not depend on which accessors ran before
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    # a covariance that is not positive definite is reported by posterior() itself, as `cholesky` throws in the reference
    with pytest.raises(P.PosDefException):
        P.posterior(Fp(xp, np.float32(-5.0)), y)


def test_one_output_type_rule_across_the_operator_surface():
    """Results come back in the element type of the inputs they were asked at (test/gp/util.jl:76-88 checks rand and
    logpdf of Float32 models for it; the advisor's round-2 finding: var / cov(fx, gx) / posterior accessors returned
    Float64 for Float32 models and the logpdf type depended on the noise kind): Float32 everywhere for an all-Float32
    model, Float64 as soon as one input is Float64 -- whichever arithmetic ran underneath."""
    rng = np.random.default_rng(21)
    f = P.atomic(P.GP(P.Matern32Kernel()), P.GPC())
    x32 = P.ColVecs(np.asfortranarray(rng.standard_normal((2, 90)).astype(np.float32)))
    z32 = P.ColVecs(np.asfortranarray(rng.standard_normal((2, 40)).astype(np.float32)))
    x64 = P.ColVecs(np.asfortranarray(x32.X.astype(np.float64)))
    y32 = rng.standard_normal(90).astype(np.float32)
    fx, gx = f(x32, np.float32(0.1)), f(z32, np.float32(0.2))
    S = 0.1 * np.eye(90) + 0.01
    f32 = np.float32
    assert isinstance(P.logpdf(fx, y32), f32)
    assert isinstance(P.logpdf(f(x32, S), y32), f32)                                   # dense Sigma_y: fp64 arithmetic
    assert P.logpdf(fx, np.stack([y32, y32], axis=1)).dtype == f32                     # matrix Y: fp64 arithmetic
    assert P.mean(fx).dtype == f32 and P.var(fx).dtype == f32 and P.cov(fx).dtype == f32
    assert P.cov(fx, gx).dtype == f32 and P.cov(fx, gx).shape == (90, 40)
    m, v = P.mean_and_var(fx)
    assert m.dtype == f32 and v.dtype == f32
    assert P.rand(np.random.default_rng(1), fx).dtype == f32
    post = P.posterior(fx, y32)
    assert post.mean(z32).dtype == f32 and post.var(z32).dtype == f32 and post.cov(z32).dtype == f32
    assert post.cov(z32, x32).dtype == f32 and post.cov(z32, x32).shape == (40, 90)
    mm, cc = post.mean_and_cov(z32)
    assert mm.dtype == f32 and cc.dtype == f32
    # one Float64 input anywhere: Float64 results
    assert isinstance(P.logpdf(f(x64, 0.1), y32.astype(np.float64)), float)
    assert P.var(f(x64, 0.1)).dtype == np.float64 and P.cov(fx, f(x64, 0.1)).dtype == np.float64
    assert post.mean(P.ColVecs(np.asfortranarray(z32.X.astype(np.float64)))).dtype == np.float64
    # and the numbers are the fp64 ones, rounded
    ref = P.posterior(f(x64, float(f32(0.1))), y32.astype(np.float64))
    assert np.max(np.abs(post.cov(z32) - ref.cov(P.ColVecs(np.asfortranarray(z32.X.astype(np.float64)))))) <= 1e-5


def test_lds_dma_update_kernel_gives_the_bits_of_the_register_staged_one(monkeypatch):
    """Round 4: the fp32 trailing update moved to the fp64 kernel's operand path (global -> LDS directly, csrc/f32.hip:
    gemm_nt_f32_dma_kernel).  Same MFMA, same k order, same C -= epilogue: the factor -- hence logpdf and the posterior
    moments -- must not change by a bit against the round-3 kernel (SGP_F32_DMA=0), on a size with several panels, a
    ragged last tile and the bordered rows."""
    rng = np.random.default_rng(5)
    N, D = 3300, 3
    X = np.asfortranarray(rng.standard_normal((D, N)).astype(np.float32))
    y = rng.standard_normal(N).astype(np.float32)
    Xs = np.asfortranarray(rng.standard_normal((D, 200)).astype(np.float32))
    f = P.atomic(P.GP(P.Matern52Kernel()), P.GPC())

    def run():
        fx = f(P.ColVecs(X), np.float32(0.3))
        return P.logpdf(fx, y), P.posterior_mean_and_var_f32(fx, y, P.ColVecs(Xs))

    monkeypatch.setenv("SGP_F32_DMA", "0")
    lp0, (m0, v0) = run()
    monkeypatch.setenv("SGP_F32_DMA", "1")
    lp1, (m1, v1) = run()
    assert lp0 == lp1
    assert np.array_equal(m0, m1) and np.array_equal(v0, v1)


@pytest.mark.parametrize("n", [384, 1536, 1400])
def test_logpdf_f32_skips_the_structural_zeros_and_keeps_its_bits(monkeypatch, n):
    """Round 6: the fp32 factorisation takes the fp64 driver's tile pattern of the factor (csrc/f32.hip: sgp_logpdf_f32 ->
    drv_sz_pattern; gemm_nt_f32_dma_kernel and panel_solve_f32_kernel skip / trim by it).  f3 = f1 + f2 observed at all
    three, ordered f1, f2, f3: the (f2, f1) block of the factor is zero.  The skipped products are exact zeros, so logpdf must
    not change by a bit against SGP_STRUCT_ZEROS=0 -- aligned blocks, several outer panels, ragged blocks (1400: block
    boundaries inside tiles) -- while the work counter shows the skipping; the value stays the oracle's at fp32 tolerance."""
    from test_gpu_fused_potrf import _with_ctx
    rng = np.random.default_rng(n)
    xs = {k: np.asfortranarray(rng.standard_normal((2, n)).astype(np.float32)) for k in ("f1", "f2", "f3")}
    y = rng.standard_normal(3 * n).astype(np.float32)
    Fp = P.gppp_sum_model()

    def logpdf():
        x = P.BlockData([P.GPPPInput(k, P.ColVecs(xs[k])) for k in ("f1", "f2", "f3")])
        return P.logpdf(Fp(x, np.float32(0.1)), y)

    out = {}
    for sz in (0, 1):
        monkeypatch.setenv("SGP_STRUCT_ZEROS", str(sz))
        ctx = P.lib.Context(0)
        out[sz] = [_with_ctx(ctx, logpdf) for _ in range(2)]
        e, d = ctx.factor_work()
        assert (e < 0.75 * d) if sz else (e == d), (sz, e, d)
        ctx.close()
    assert isinstance(out[1][0], np.float32)
    assert out[0][0] == out[0][1] == out[1][0] == out[1][1], out
    from oracle import reference_model as orm
    xo = ost.BlockData([ost.GPPPInput(k, okf.ColVecs(xs[k].astype(np.float64))) for k in ("f1", "f2", "f3")])
    ref = oagp.logpdf(orm.gppp_sum()(xo, float(np.float32(0.1))), y.astype(np.float64))
    assert abs(float(out[1][0]) - ref) <= 2e-4 * max(1.0, abs(ref)), (out[1][0], ref)
