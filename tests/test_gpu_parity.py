"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI via the
product's host mirror, against the CPU oracle on identical seeded inputs.

Tolerances (fp64): north star demands logpdf and posterior mean/var within 1e-8 relative; the
tests hold 1e-10 where conditioning allows and say so where it does not.  Exact identities the
reference's tests pin with `==` are asserted bit-exact here as well."""
import numpy as np
import pytest

import models
import oracle.abstractgps as oagp
import oracle.kernelfunctions as okf
import oracle.stheno as ost
import stheno_jl_amd as P

pytestmark = pytest.mark.gpu

REL = 1e-10


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def both(recipe):
    fo, go = recipe(models.oracle_api())
    fp, gp = recipe(models.product_api())
    return ost.GPPP(fo, go), P.GPPP(fp, gp), fo, fp


def blockdata(names, xs, colvecs):
    wrap_o = (lambda x: okf.ColVecs(x)) if colvecs else (lambda x: x)
    wrap_p = (lambda x: P.ColVecs(x)) if colvecs else (lambda x: x)
    xo = ost.BlockData([ost.GPPPInput(k, wrap_o(x)) for k, x in zip(names, xs)])
    xp = P.BlockData([P.GPPPInput(k, wrap_p(x)) for k, x in zip(names, xs)])
    return xo, xp


def test_library_is_native_and_loaded():
    import ctypes
    lib = P.lib.load()
    assert lib.sgp_abi_version() == 1
    ctx = P.lib.default_context()
    assert ctx.handle
    # the loaded object is the in-tree HIP library
    with open("/proc/self/maps") as fh:
        assert any("stheno.jl_amd/csrc/libsthenomi.so" in ln for ln in fh)
    tf, err = ctypes.c_double(), ctypes.c_double()
    P.lib.check(ctx.bench.sgp_bench_mfma_f64(ctx.handle, 50, ctypes.byref(tf), ctypes.byref(err)))
    assert err.value == 0.0  # documented MFMA lane maps hold on this device


# ---- covariance assembly ---------------------------------------------------------------------
@pytest.mark.parametrize("recipe", models.RECIPES_1D, ids=lambda r: r.__name__)
def test_cov_mean_var_blockdata_1d(recipe):
    rng = np.random.default_rng(123456)
    Fo, Fp, fo, fp = both(recipe)
    names = list(fo)
    xs = [rng.standard_normal(40 + 37 * i) for i in range(len(names))]
    xo, xp = blockdata(names, xs, False)
    Ko = Fo.cov(xo)
    Kp = P.prior_cov(Fp, xp)
    # 1e-11: the oracle's GEMM-trick distances carry O(eps |x|^2) noise that sqrt() amplifies for
    # the kernels that are not smooth at 0 (Matern-1/2 in `warped`), SURVEY.md App. A.1 caveat
    assert np.abs(Kp - Ko).max() < 1e-11
    assert np.array_equal(Kp, Kp.T)                      # exactly symmetric (lower triangle mirrored)
    assert np.all(Kp[np.abs(Ko) == 0.0] == 0.0)          # independent blocks are exact zeros
    np.testing.assert_allclose(P.prior_var(Fp, xp), Fo.var(xo), rtol=0, atol=1e-12)
    assert np.array_equal(P.prior_var(Fp, xp), np.diag(Kp))  # var == diag(cov) bit for bit
    np.testing.assert_allclose(P.prior_mean(Fp, xp), Fo.mean(xo), rtol=0, atol=1e-13)
    # cross covariance between two different input collections
    ys = [rng.standard_normal(5 + 3 * i) for i in range(len(names))]
    yo, yp = blockdata(names[::-1], ys, False)
    np.testing.assert_allclose(P.prior_cov(Fp, xp, yp), Fo.cov(xo, yo), rtol=0, atol=1e-12)
    np.testing.assert_allclose(P.prior_cov(Fp, yp, xp), P.prior_cov(Fp, xp, yp).T, rtol=0, atol=1e-15)


@pytest.mark.parametrize("recipe", [models.gppp_docstring, models.correlated_sums, models.scaled,
                                    models.composite_kernels], ids=lambda r: r.__name__)
@pytest.mark.parametrize("D", [1, 2, 3, 8, 17])
def test_cov_colvecs(recipe, D):
    rng = np.random.default_rng(7 + D)
    Fo, Fp, fo, fp = both(recipe)
    names = list(fo)[:4]
    xs = [np.asfortranarray(rng.standard_normal((D, 60 + 71 * i))) for i in range(len(names))]
    xo, xp = blockdata(names, xs, True)
    Ko, Kp = Fo.cov(xo), P.prior_cov(Fp, xp)
    assert np.abs(Kp - Ko).max() < 2e-12 * max(1.0, np.abs(Ko).max())


@pytest.mark.parametrize("D", [33, 64, 65, 100, 257])
def test_cov_and_logpdf_high_dimensional_inputs(D):
    """ColVecs of any dimension (KernelFunctions.kernelmatrix [EXT] has no limit): D <= 64 runs on the templated
    one-row kernel, D > 64 on assemble_bigd_kernel (dimension walked in chunks of 16); several terms per block
    pair accumulate launch by launch.  Inputs scaled by 1 / sqrt(D) so that distances stay O(1)."""
    import np_terms
    rng = np.random.default_rng(500 + D)
    Fo, Fp, fo, fp = both(models.correlated_sums)
    names = list(fo)[:3]
    xs = [np.asfortranarray(rng.standard_normal((D, n)) / np.sqrt(D)) for n in (140, 1, 203)][:len(names)]
    xo, xp = blockdata(names, xs, True)
    Ko, Kp = Fo.cov(xo), P.prior_cov(Fp, xp)
    assert np.abs(Kp - Ko).max() < 1e-11 * max(1.0, np.abs(Ko).max())
    spec, _, _ = P.build_spec(Fp, xp)
    Kd = np_terms.dense_from_spec(spec)                      # direct differences, like the device
    Kd = np.tril(Kd) + np.tril(Kd, -1).T
    assert np.abs(Kp - Kd).max() < 2e-13 * max(1.0, np.abs(Kd).max())
    assert np.array_equal(P.prior_var(Fp, xp), np.diag(Kp))
    N = sum(x.shape[1] for x in xs)
    y = rng.standard_normal(N)
    lo, lp = oagp.logpdf(Fo(xo, 0.2), y), P.logpdf(Fp(xp, 0.2), y)
    assert abs(lp - lo) <= REL * abs(lo)
    # cross-covariance and the posterior path use the same assembly
    xq = [np.asfortranarray(rng.standard_normal((D, 37)) / np.sqrt(D))]
    xqo, xqp = blockdata(names[:1], xq, True)
    po, pp = oagp.posterior(Fo(xo, 0.2), y), P.posterior(Fp(xp, 0.2), y)
    assert rel(pp.mean(xqp), po.mean(xqo)) < 1e-9 and np.max(np.abs(pp.var(xqp) - po.var(xqo))) < 1e-9


@pytest.mark.parametrize("D", [12, 16])
def test_cov_many_terms_per_block_pair_accumulate_path(D):
    """More terms per block pair than one assembly launch can stage in LDS (D = 16: one term per launch): the
    following launches accumulate into the tile (read-modify-write path of assemble_block2_kernel), with row /
    column scales and ragged, tile-misaligned blocks."""
    rng = np.random.default_rng(40 + D)

    def recipe(api):
        gpc = api.GPC()
        k = api.KernelSum([api.SEKernel(), api.with_lengthscale(api.Matern52Kernel(), 2.0),
                           api.ScaledKernel(api.Matern32Kernel(), 0.3), api.ConstantKernel(0.2)])
        f1 = api.atomic(api.GP(k), gpc)
        # (no Matern-1/2 here: the oracle's GEMM-trick distances make exp(-sqrt(d2)) 1e-8 off at coincident points
        # of a cross-covariance, SURVEY.md App. A.1 -- the device's direct differences are exact there)
        f2 = api.atomic(api.GP(api.Matern32Kernel()), gpc)
        return {"f1": f1, "f2": f2, "s": models._sumsin * f1 + 0.5 * f2 + api.stretch(f1, 0.7)}, gpc

    Fo, Fp, fo, fp = both(recipe)
    names = ["s", "f1", "s"]
    xs = [np.asfortranarray(rng.standard_normal((D, n)) / np.sqrt(D)) for n in (150, 131, 77)]
    xo, xp = blockdata(names, xs, True)
    Ko, Kp = Fo.cov(xo), P.prior_cov(Fp, xp)
    assert np.abs(Kp - Ko).max() < 5e-12 * max(1.0, np.abs(Ko).max())
    assert np.array_equal(Kp, Kp.T)
    y = rng.standard_normal(len(xp))
    lo, lp = oagp.logpdf(Fo(xo, 0.1), y), P.logpdf(Fp(xp, 0.1), y)
    assert abs(lp - lo) <= 1e-10 * abs(lo)


def test_warps_colvecs():
    rng = np.random.default_rng(99)
    Fo, Fp, fo, fp = both(models.warped_colvecs)
    X = np.asfortranarray(rng.standard_normal((3, 150)))
    for name in ["f", "g", "fs", "fv", "fA", "fsh", "fsel", "fsel1", "add"]:
        Ko = fo[name].cov(okf.ColVecs(X))
        Kp = P.prior_cov(fp[name], P.ColVecs(X))
        assert np.abs(Kp - Ko).max() < 1e-12, name


def test_exact_identities_from_reference_tests():
    """`==` pins: test/gp/atomic_gp.jl:15,33,34; test/affine_transformations/compose.jl:53-54,73-74."""
    rng = np.random.default_rng(3)
    gpc = P.GPC()
    f1 = P.atomic(P.GP(P.SEKernel()), gpc)
    f2 = P.atomic(P.GP(5, P.SEKernel()), gpc)
    x, x2 = rng.standard_normal(5), rng.standard_normal(6)
    assert np.array_equal(P.prior_var(f1, x), np.ones(5))                       # SE diagonal == 1
    K12 = P.cov(f1(x), f2(x2))
    assert K12.shape == (5, 6) and np.array_equal(K12, np.zeros((5, 6)))         # independent atoms
    lam = 0.51
    g = P.stretch(f1, lam)
    xs = rng.standard_normal(1)
    assert P.cov(f1(lam * xs), g(xs))[0, 0] == 1.0                                # cov(f, stretch(f,l), [l x],[x]) == 1
    Dn = 11
    Xc = rng.standard_normal((Dn, 1))
    assert P.cov(f1(P.ColVecs(lam * Xc)), g(P.ColVecs(Xc)))[0, 0] == 1.0
    fg = P.compose(f1, np.cos)
    assert np.array_equal(P.prior_cov(fg, x), P.prior_cov(f1, np.cos(x)))        # compose.jl:15-16
    # cov(f, x) == kernelmatrix(k, x): a one-atom tree is exactly the leaf kernel matrix
    np.testing.assert_allclose(P.prior_cov(f1, x), okf.kernelmatrix(okf.SEKernel(), x, faithful=False),
                               rtol=0, atol=2e-16)


# ---- logpdf ---------------------------------------------------------------------------------------
SIZES = [1, 2, 127, 128, 129, 300, 1000]


@pytest.mark.parametrize("N", SIZES)
@pytest.mark.parametrize("kind", ["se", "matern52", "matern32", "matern12"])
def test_logpdf_single_gp(N, kind):
    rng = np.random.default_rng(N * 7 + len(kind))
    D = 3
    ko = {"se": okf.SEKernel, "matern52": okf.Matern52Kernel, "matern32": okf.Matern32Kernel,
          "matern12": okf.Matern12Kernel}[kind]()
    kp = {"se": P.SEKernel, "matern52": P.Matern52Kernel, "matern32": P.Matern32Kernel,
          "matern12": P.Matern12Kernel}[kind]()
    fo = ost.atomic(oagp.GP(0.3, ko), ost.GPC())
    fp = P.atomic(P.GP(0.3, kp), P.GPC())
    X = np.asfortranarray(rng.standard_normal((D, N)))
    y = rng.standard_normal(N)
    Y = rng.standard_normal((N, 3))
    for noise in (0.1, 0.05 + rng.random(N)):
        lo = oagp.logpdf(fo(okf.ColVecs(X), noise), y)
        lp = P.logpdf(fp(P.ColVecs(X), noise), y)
        assert abs(lp - lo) <= REL * abs(lo), (lp, lo)
        Lo = oagp.logpdf(fo(okf.ColVecs(X), noise), Y)
        Lp = P.logpdf(fp(P.ColVecs(X), noise), Y)
        assert Lp.shape == (3,) and rel(Lp, Lo) < REL
    if N <= 300:
        A = rng.standard_normal((N, N))
        S = A @ A.T / N + 0.1 * np.eye(N)
        lo = oagp.logpdf(fo(okf.ColVecs(X), S), y)
        lp = P.logpdf(fp(P.ColVecs(X), S), y)
        assert abs(lp - lo) <= REL * abs(lo)


@pytest.mark.parametrize("recipe", [models.gppp_docstring, models.toy_gppp, models.correlated_sums,
                                    models.composite_kernels], ids=lambda r: r.__name__)
def test_logpdf_rand_posterior_gppp(recipe):
    rng = np.random.default_rng(42)
    Fo, Fp, fo, fp = both(recipe)
    names = list(fo)[:3]
    xs = [rng.standard_normal(n) for n in (211, 95, 160)][:len(names)]
    xo, xp = blockdata(names, xs, False)
    N = sum(len(x) for x in xs)
    y = rng.standard_normal(N)
    s2 = 0.2
    lo, lp = oagp.logpdf(Fo(xo, s2), y), P.logpdf(Fp(xp, s2), y)
    assert abs(lp - lo) <= REL * abs(lo)
    # rand: same Z -> same sample (deterministic kernels; summation order differs from LAPACK)
    Z = rng.standard_normal((N, 4))
    Ro = oagp.rand(Fo(xo, s2), Z)
    Rp = P.rand(None, Fp(xp, s2), 4, Z=Z)
    assert rel(Rp, Ro) < 1e-11
    Rp2 = P.rand(None, Fp(xp, s2), 4, Z=Z)
    assert np.array_equal(Rp, Rp2)                       # bit-identical run to run
    # posterior at new inputs of two processes
    po, pp = oagp.posterior(Fo(xo, s2), y), P.posterior(Fp(xp, s2), y)
    assert rel(pp.alpha, po.alpha) < 1e-9
    ts = [np.linspace(-2, 2, 33), rng.standard_normal(20)]
    to, tp = blockdata([names[-1], names[0]], ts, False)
    mo, vo = po.mean_and_var(to)
    mp, vp = pp.mean_and_var(tp)
    assert rel(mp, mo) < REL and np.abs(vp - vo).max() < 1e-10
    Co, Cp = po.cov(to), pp.cov(tp)
    assert np.abs(Cp - Co).max() < 1e-10
    assert np.abs(np.diag(Cp) - vp).max() < 1e-12
    # a posterior FiniteGP is again a FiniteGP: logpdf / rand / marginals on top of it
    yt = rng.standard_normal(len(to))
    assert abs(P.logpdf(pp(tp, 0.3), yt) - oagp.logpdf(po(to, 0.3), yt)) <= 1e-9 * abs(oagp.logpdf(po(to, 0.3), yt))
    mg = P.marginals(pp(tp, 0.3))
    m_o, s_o = oagp.marginals(po(to, 0.3))
    assert rel([g.mu for g in mg], m_o) < REL and rel([g.sigma for g in mg], s_o) < REL


def test_posterior_external_consistency():
    """test/gaussian_process_probabilistic_programme.jl:28-42: GPPP == manual construction."""
    rng = np.random.default_rng(5)
    fp, gp = models.toy_gppp(models.product_api())
    F = P.GPPP(fp, gp)
    x0, x1 = rng.standard_normal(4), rng.standard_normal(3)
    assert np.array_equal(P.prior_mean(fp["f1"], x0), P.prior_mean(F, P.GPPPInput("f1", x0)))
    assert np.array_equal(P.prior_cov(fp["f3"], x1), P.prior_cov(F, P.GPPPInput("f3", x1)))
    assert np.array_equal(P.cov(fp["f1"](x0), fp["f3"](x1)), P.prior_cov(F, P.GPPPInput("f1", x0), P.GPPPInput("f3", x1)))
    y = P.rand(np.random.default_rng(1), F(P.GPPPInput("f3", x1)))
    a = P.posterior(fp["f3"](x1), y)(x1)
    b = P.posterior(F(P.GPPPInput("f3", x1)), y)(P.GPPPInput("f3", x1))
    assert np.array_equal(P.cov(a), P.cov(b))


def test_sequential_conditioning_posterior_of_posterior():
    """AbstractGPs allows posterior(f_post(x2, s2), y2); the oracle does it as a second conditioning on the
    first posterior's mean / cov, the product as ONE conditioning of the prior on the stacked data."""
    rng = np.random.default_rng(31)
    Fo, Fp, fo, fp = both(models.gppp_docstring)
    D = 2
    x1 = np.asfortranarray(rng.standard_normal((D, 70)))
    x2 = np.asfortranarray(rng.standard_normal((D, 40)))
    xs = np.asfortranarray(rng.standard_normal((D, 25)))
    y1, y2 = rng.standard_normal(70), rng.standard_normal(40)
    po1 = oagp.posterior(Fo(ost.GPPPInput("f3", okf.ColVecs(x1)), 0.1), y1)
    po2 = oagp.posterior(po1(ost.GPPPInput("f1", okf.ColVecs(x2)), 0.3), y2)
    pp1 = P.posterior(Fp(P.GPPPInput("f3", P.ColVecs(x1)), 0.1), y1)
    pp2 = P.posterior(pp1(P.GPPPInput("f1", P.ColVecs(x2)), 0.3), y2)
    for name in ("f1", "f2", "f3"):
        mo, vo = po2.mean_and_var(ost.GPPPInput(name, okf.ColVecs(xs)))
        mp, vp = pp2.mean_and_var(P.GPPPInput(name, P.ColVecs(xs)))
        assert np.max(np.abs(mp - mo)) < 1e-9 and np.max(np.abs(vp - vo)) < 1e-9, name
    # plain (non-GPPP) prior, equal scalar noises
    f_o, f_p = ost.atomic(oagp.GP(okf.Matern32Kernel()), ost.GPC()), P.atomic(P.GP(P.Matern32Kernel()), P.GPC())
    a, b, c = rng.standard_normal(30), rng.standard_normal(20), rng.standard_normal(9)
    ya, yb = rng.standard_normal(30), rng.standard_normal(20)
    qo = oagp.posterior(oagp.posterior(f_o(a, 0.2), ya)(b, 0.2), yb)
    qp = P.posterior(P.posterior(f_p(a, 0.2), ya)(b, 0.2), yb)
    assert np.max(np.abs(qp.mean(c) - qo.mean(c))) < 1e-9
    assert np.max(np.abs(qp.cov(c) - qo.cov(c))) < 1e-9


def test_non_positive_definite_raises_posdef():
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    x = np.zeros(10)                       # rank-one covariance, negative "noise"
    with pytest.raises(P.PosDefException) as ei:
        P.logpdf(f(x, -0.5), np.zeros(10))
    assert ei.value.info == 2              # LAPACK potrf convention: first failing leading minor
    with pytest.raises(oagp.PosDefException):
        oagp.logpdf(ost.atomic(oagp.GP(okf.SEKernel()), ost.GPC())(x, -0.5), np.zeros(10))


def test_rand_statistics():
    """test/gp/util.jl:36-47: S = 100000 samples, N = 10, D = 2."""
    rng = np.random.default_rng(123456)
    X = P.ColVecs(rng.standard_normal((2, 10)))
    fx = P.atomic(P.GP(1, P.SEKernel()), P.GPC())(X, 1e-12)
    S = 100_000
    fh = P.rand(rng, fx, S)
    assert fh.shape == (10, S)
    assert np.abs(fh.mean(1) - P.mean(fx)).max() < 1e-2
    Sig = (fh - P.mean(fx)[:, None]) @ (fh - P.mean(fx)[:, None]).T / S
    assert np.mean(np.abs(Sig - P.cov(fx))) < 1e-2
    assert P.rand(rng, fx).shape == (10,)


def test_rand_sum_model_consistency():
    """additive sample check, @gppp docstring (gppp.jl:150-160): f1 + f2 ~= f3 at s2 = 1e-12."""
    rng = np.random.default_rng(0)
    F = P.gppp_sum_model()
    xl = rng.standard_normal(5)
    x = P.BlockData([P.GPPPInput("f1", xl), P.GPPPInput("f2", xl), P.GPPPInput("f3", xl)])
    y = P.rand(rng, F(x, 1e-12))
    a, b, c = P.split(x, y)
    np.testing.assert_allclose(a + b, c, rtol=1e-4, atol=1e-4)


# ---- sparse / VFE --------------------------------------------------------------------------------
def test_sparse_finite_gp_reference_properties():
    """test/gp/sparse_finite_gp.jl:1-42."""
    x = np.arange(0.0, 10.0001, 0.1)
    xu = np.arange(0.0, 10.5, 1.0)
    sig, sigu = 1.0, 1e-3
    fo = ost.atomic(oagp.GP(okf.Matern32Kernel()), ost.GPC())
    fp = P.atomic(P.GP(P.Matern32Kernel()), P.GPC())
    fx = fp(x, sig)
    fxu = P.SparseFiniteGP(fp(x, sig), fp(xu, sigu))
    assert len(fxu) == len(x)
    y = P.rand(np.random.default_rng(12345), fxu)
    assert np.array_equal(y, P.rand(np.random.default_rng(12345), fx))       # samples the dense fobs
    e1 = P.elbo(fxu, y)
    assert e1 == P.logpdf(fxu, y) == P.elbo(P.VFE(fxu.finducing), fxu.fobs, y)
    e_o = oagp.elbo(oagp.VFE(fo(xu, sigu)), fo(x, sig), y)
    assert abs(e1 - e_o) <= 1e-9 * abs(e_o)
    yy = P.rand(np.random.default_rng(3), fxu, 10)
    assert np.all(P.logpdf(fx, yy) > P.logpdf(fxu, yy))                       # ELBO is a lower bound
    p1 = P.posterior(P.VFE(fxu.finducing), fxu.fobs, y)      # the 3-argument AbstractGPs form
    p2 = P.posterior(fxu, y)
    m1, v1 = p1.mean_and_var(x)
    m2, v2 = p2.mean_and_var(x)
    assert np.array_equal(m1, m2) and np.array_equal(v1, v2)
    po = oagp.posterior_vfe(oagp.VFE(fo(xu, sigu)), fo(x, sig), y)
    assert rel(m1, po.mean(x)) < 1e-8 and np.abs(v1 - po.var(x)).max() < 1e-8
    assert np.abs(p1.cov(x[:40]) - po.cov(x[:40])).max() < 1e-8
    with pytest.raises(RuntimeError):
        P.sparse_cov(fxu)


def test_elbo_equals_logpdf_when_z_is_x():
    """README.md:71-78 (not CI'd in the reference): elbo -> logpdf for Z = X, tiny jitter."""
    rng = np.random.default_rng(8)
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    x = np.sort(rng.uniform(-5, 5, 60))
    y = rng.standard_normal(60)
    lp = P.logpdf(f(x, 0.1), y)
    el = P.elbo(P.VFE(f(x, 1e-9)), f(x, 0.1), y)
    assert abs(lp - el) < 1e-5 * abs(lp)


@pytest.mark.parametrize("N,M,D", [(777, 130, 2), (3000, 256, 4)])
def test_elbo_gppp_with_diag_noise(N, M, D):
    rng = np.random.default_rng(N)
    Fo, Fp, fo, fp = both(models.gppp_docstring)
    X = np.asfortranarray(rng.standard_normal((D, N)))
    Z = np.asfortranarray(X[:, rng.permutation(N)[:M]])
    y = rng.standard_normal(N)
    noise = 0.05 + rng.random(N)
    eo = oagp.elbo(oagp.VFE(Fo(ost.GPPPInput("f3", okf.ColVecs(Z)), 1e-6)), Fo(ost.GPPPInput("f3", okf.ColVecs(X)), noise), y)
    ep = P.elbo(P.VFE(Fp(P.GPPPInput("f3", P.ColVecs(Z)), 1e-6)), Fp(P.GPPPInput("f3", P.ColVecs(X)), noise), y)
    assert abs(ep - eo) <= 1e-9 * abs(eo)


# ---- full-size, size-independent properties -----------------------------------------------------------
# (BASELINE-size parity lives in tests/test_gpu_baseline_golden.py: CPU known-answer values of
# c1..c5, n4k and the north-star target model.  The former N = 16384 round-trip / scaling test was
# self-consistent by construction -- a consistent error in L cancels in y = m + L z -> L^-1 (y - m)
# -- and has been replaced by it.)


def test_golden_sklearn_vectors():
    """Committed golden vectors (tests/golden/make_golden.py): an independent third
    implementation (scikit-learn) pins both the oracle and the HIP path."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sklearn_gpr.json")))
    for case in g["cases"]:
        X = np.asfortranarray(np.array(case["X"]).T)       # stored N x D
        y = np.array(case["y"])
        kp = {"se": P.SEKernel, "matern52": P.Matern52Kernel, "matern32": P.Matern32Kernel}[case["kernel"]]()
        f = P.stretch(P.atomic(P.GP(kp), P.GPC()), 1.0 / case["lengthscale"])
        lp = P.logpdf(f(P.ColVecs(X), case["sigma2"]), y)
        assert abs(lp - case["lml"]) <= 1e-9 * abs(case["lml"])
        post = P.posterior(f(P.ColVecs(X), case["sigma2"]), y)
        Xs = P.ColVecs(np.asfortranarray(np.array(case["Xs"]).T))
        m, v = post.mean_and_var(Xs)
        assert rel(m, case["mean"]) < 1e-8 and np.abs(np.sqrt(v) - np.array(case["std"])).max() < 1e-8


# ---- edge cases: empty / single-point / ragged blocks, many right-hand sides -----------------------
def test_ragged_and_empty_blocks():
    rng = np.random.default_rng(11)
    Fo, Fp, fo, fp = both(models.gppp_docstring)
    xs = [rng.standard_normal(1), np.zeros(0), rng.standard_normal(257)]
    xo, xp = blockdata(["f1", "f2", "f3"], xs, False)
    N = 258
    y = rng.standard_normal(N)
    lo, lp = oagp.logpdf(Fo(xo, 0.3), y), P.logpdf(Fp(xp, 0.3), y)
    assert abs(lp - lo) <= REL * abs(lo)
    Ko, Kp = Fo.cov(xo), P.prior_cov(Fp, xp)
    assert Kp.shape == (N, N) and np.abs(Kp - Ko).max() < 1e-12
    post_o, post_p = oagp.posterior(Fo(xo, 0.3), y), P.posterior(Fp(xp, 0.3), y)
    t = rng.standard_normal(3)
    mo, vo = post_o.mean_and_var(ost.GPPPInput("f2", t))
    mp, vp = post_p.mean_and_var(P.GPPPInput("f2", t))
    assert rel(mp, mo) < 1e-9 and np.abs(vp - vo).max() < 1e-10


@pytest.mark.parametrize("S", [1, 5, 130])
def test_many_right_hand_sides(S):
    rng = np.random.default_rng(S)
    N = 333
    fo = ost.atomic(oagp.GP(okf.Matern32Kernel()), ost.GPC())
    fp = P.atomic(P.GP(P.Matern32Kernel()), P.GPC())
    x = rng.standard_normal(N)
    Y = rng.standard_normal((N, S))
    Lo, Lp = oagp.logpdf(fo(x, 0.2), Y), P.logpdf(fp(x, 0.2), Y)
    assert Lp.shape == (S,) and rel(Lp, Lo) < REL
    Z = rng.standard_normal((N, S))
    assert rel(P.rand(None, fp(x, 0.2), S, Z=Z), oagp.rand(fo(x, 0.2), Z)) < 1e-11


def test_input_dimension_limit_is_reported():
    """(The name is history: there is no limit left to report.)  Assembly (cov, logpdf, posterior, rand, elbo) takes ColVecs
    of any dimension; so do the term gradients (round 4: grad_block_bigd_kernel walks the dimension in chunks of 16 beyond
    64, one term per launch; they used to stop at 64 with an error) and the input gradients (round 3:
    grad_inputs_bigd_kernel; they used to stop at 16).  Term gradients at D = 65 and 130 against the oracle's cotangent
    contraction and against central differences of the logpdf in the variance and the lengthscale; input gradients against
    central differences at D = 17 and 40, incl. a function-valued scale (the row-scale sums of the same kernel)."""
    f = P.atomic(P.GP(P.SEKernel()), P.GPC())
    rng = np.random.default_rng(0)
    for D in (64, 65):
        X = P.ColVecs(rng.standard_normal((D, 40)) / np.sqrt(D))
        K = P.prior_cov(f, X)
        assert np.abs(K - okf.kernelmatrix(okf.SEKernel(), okf.ColVecs(X.X), faithful=False)).max() < 1e-13
    for D in (65, 130):
        n = 300                                   # three row tiles x three column tiles, ragged last tile
        Xd = np.asfortranarray(rng.standard_normal((D, n)) / np.sqrt(D))
        yd = rng.standard_normal(n)

        def model(v, l):
            return np.sqrt(v) * P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 1.0 / l)

        v, l, h = 1.3, 0.7, 1e-5
        g = P.logpdf_and_gradient(model(v, l)(P.ColVecs(Xd), 0.2), yd)
        (term,) = g["terms"]
        lp = lambda vv, ll: P.logpdf(model(vv, ll)(P.ColVecs(Xd), 0.2), yd)
        fd_v = (lp(v + h, l) - lp(v - h, l)) / (2 * h)
        fd_l = (lp(v, l + h) - lp(v, l - h)) / (2 * h)
        assert abs(term["d_coef"] - fd_v) <= 1e-6 * max(1.0, abs(fd_v)), (D, term["d_coef"], fd_v)
        d_l = -(1.0 / l) * term["d_inscale"]
        assert abs(d_l - fd_l) <= 1e-6 * max(1.0, abs(fd_l)), (D, d_l, fd_l)
        gc, gs = g["_raw"]
        Ko = v * okf.kernelmatrix(okf.Matern52Kernel(), okf.ColVecs(Xd / l), faithful=False) + 0.2 * np.eye(n)
        Ci = np.linalg.inv(Ko)
        al = Ci @ yd
        Gm = 0.5 * (np.outer(al, al) - Ci)
        (ec, es), = _oracle_term_grads(g["_spec"], Gm)
        assert abs(gc[0] - ec) <= 1e-8 * max(1.0, abs(ec)) and abs(gs[0] - es) <= 2e-6 * max(1.0, abs(es))
    y60 = rng.standard_normal(60)
    for D in (17, 40):
        X = np.asfortranarray(rng.standard_normal((D, 60)) / np.sqrt(D))
        sig = lambda pt: 1.0 + 0.3 * float(np.sin(np.sum(pt)))
        for proc in (f, sig * f):
            lp = lambda Xq: P.logpdf(proc(P.ColVecs(np.asfortranarray(Xq)), 0.1), y60)
            g = P.logpdf_and_gradient(proc(P.ColVecs(X), 0.1), y60, inputs=True, scales=proc is not f)
            assert abs(g["logpdf"] - lp(X)) <= 1e-12 * abs(g["logpdf"])
            gx = g["inputs"][0]                 # d logpdf / d (the points the kernel reads), sigma(x) held fixed
            assert gx.shape == (D, 60)
            if proc is f:
                for (d, i) in ((0, 0), (D - 1, 59), (D // 2, 17)):
                    h = 1e-6
                    Xp, Xm = X.copy(), X.copy()
                    Xp[d, i] += h
                    Xm[d, i] -= h
                    fd = (lp(Xp) - lp(Xm)) / (2 * h)
                    assert abs(gx[d, i] - fd) <= 1e-5 * max(1.0, abs(fd)), (D, d, i, gx[d, i], fd)
            else:
                # total derivative w.r.t. one coordinate = through the kernel's points + through sigma at that point
                (rec,) = g["scales"]
                for (d, i) in ((1, 3), (D - 2, 41)):
                    h = 1e-6
                    Xp, Xm = X.copy(), X.copy()
                    Xp[d, i] += h
                    Xm[d, i] -= h
                    fd = (lp(Xp) - lp(Xm)) / (2 * h)
                    dsig = 0.3 * float(np.cos(np.sum(X[:, i])))
                    tot = gx[d, i] + rec["d_values"][i] * dsig
                    assert abs(tot - fd) <= 1e-5 * max(1.0, abs(fd)), (D, d, i, tot, fd)


# ---- reverse-mode gradient of logpdf -------------------------------------------
def _oracle_term_grads(spec, G):
    """sum_ij G_ij d C_ij / d theta for every raw spec term, with NumPy (tests/np_terms.py kernels)."""
    import np_terms
    roff = np.concatenate([[0], np.cumsum(spec.row_len)])
    coff = np.concatenate([[0], np.cumsum(spec.col_len)])
    out = []
    for (I, J, kind, ri, ci, coef, param, rs, cs) in np_terms.spec_terms(spec):
        X, Y = spec.inputs[ri], spec.inputs[ci]
        d2 = ((X[:, :, None] - Y[:, None, :]) ** 2).sum(0)
        k = np_terms._kern(kind, d2, param)
        h = 1e-6
        dk = (np_terms._kern(kind, d2 * (1 + h) ** 2, param) - np_terms._kern(kind, d2 * (1 - h) ** 2, param)) / (2 * h)
        w = G[roff[I]:roff[I + 1], coff[J]:coff[J + 1]]
        if rs is not None:
            w = w * rs[:, None]
        if cs is not None:
            w = w * cs[None, :]
        out.append(((w * k).sum(), coef * (w * dk).sum()))
    return out


@pytest.mark.parametrize("recipe", [models.gppp_docstring, models.scaled, models.composite_kernels],
                         ids=lambda r: r.__name__)
def test_logpdf_gradient_terms_noise_y_mean(recipe):
    rng = np.random.default_rng(5)
    Fo, Fp, fo, fp = both(recipe)
    names = list(fo)[:3]
    D = 2
    xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (150, 70, 131)][:len(names)]
    xo, xp = blockdata(names, xs, True)
    N = sum(x.shape[1] for x in xs)
    y = rng.standard_normal(N)
    Bd = rng.standard_normal((N, 5))
    dense = 0.2 * np.eye(N) + 0.02 * Bd @ Bd.T          # f(x, S::Matrix): round 4, the gradient w.r.t. S is G itself
    for noise in (0.3, 0.1 + rng.random(N), dense):
        lp_o, alpha, G = oagp.logpdf_gradient_wrt_cov(Fo(xo, noise), y)
        g = P.logpdf_and_gradient(Fp(xp, noise), y)
        assert abs(g["logpdf"] - lp_o) <= REL * abs(lp_o)
        assert rel(g["y"], -alpha) < 1e-9 and rel(g["mean"], alpha) < 1e-9
        if np.ndim(noise) == 0:
            assert abs(g["noise"] - np.trace(G)) <= 1e-9 * max(1.0, abs(np.trace(G)))
        elif np.ndim(noise) == 2:
            assert g["noise"].shape == (N, N) and rel(g["noise"], G) < 1e-9
        else:
            assert rel(g["noise"], np.diag(G)) < 1e-9
        gc, gs = g["_raw"]
        exp = _oracle_term_grads(g["_spec"], G)
        assert len(exp) == g["_spec"].n_terms
        for t, (ec, es) in enumerate(exp):
            assert abs(gc[t] - ec) <= 1e-8 * max(1.0, abs(ec)), (t, gc[t], ec)
            assert abs(gs[t] - es) <= 2e-6 * max(1.0, abs(es)), (t, gs[t], es)   # FD reference for dk/dg


def test_logpdf_gradient_matches_finite_differences_of_hyperparameters():
    """End to end: variance, lengthscale and noise of s * stretch(GP(Matern52), 1/l) -- the
    getting_started example's parameters -- against central differences of the GPU logpdf itself."""
    rng = np.random.default_rng(9)
    X = P.ColVecs(rng.standard_normal((3, 400)))
    y = rng.standard_normal(400)

    def model(v, l):
        return np.sqrt(v) * P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 1.0 / l)

    v, l, s2 = 1.7, 0.8, 0.25
    g = P.logpdf_and_gradient(model(v, l)(X, s2), y)
    (term,) = g["terms"]
    d_v = term["d_coef"]                       # K = v * k  ->  coef == v
    d_l = -(1.0 / l) * term["d_inscale"]       # inputs X / l = g X with g = 1/l: d/dl = -(g/l) d/dg... d/dg at g=1 of (g X/l)
    h = 1e-5
    fd_v = (P.logpdf(model(v + h, l)(X, s2), y) - P.logpdf(model(v - h, l)(X, s2), y)) / (2 * h)
    fd_l = (P.logpdf(model(v, l + h)(X, s2), y) - P.logpdf(model(v, l - h)(X, s2), y)) / (2 * h)
    fd_s = (P.logpdf(model(v, l)(X, s2 + h), y) - P.logpdf(model(v, l)(X, s2 - h), y)) / (2 * h)
    assert abs(d_v - fd_v) <= 1e-6 * max(1.0, abs(fd_v))
    assert abs(d_l - fd_l) <= 1e-6 * max(1.0, abs(fd_l))
    assert abs(g["noise"] - fd_s) <= 1e-6 * max(1.0, abs(fd_s))


def test_logpdf_gradient_records_with_scale_only_differences():
    """f3 = a f1 + b (sin .* f1) on two blocks -> block pair (f3, f3') holds four terms
    that differ only in their row / column scale vectors (coefficients a^2, ab, ab, b^2).  The
    per-record d_coef (mirror pairs folded) must reproduce d logpdf / da and / db -- the records weigh
    differently in the two derivatives, so a mis-folded mirror term shows."""
    rng = np.random.default_rng(17)
    x1, x2 = rng.standard_normal(90), rng.standard_normal(70)
    y = rng.standard_normal(160)

    def sin_scale(x):
        return float(np.sum(np.sin(x)))

    def fx(a, b):
        gpc = P.GPC()
        f1 = P.atomic(P.GP(P.SEKernel()), gpc)
        F = P.GPPP({"f1": f1, "f3": a * f1 + b * (sin_scale * f1)}, gpc)
        return F(P.BlockData([P.GPPPInput("f3", x1), P.GPPPInput("f3", x2)]), 0.2)

    a, b = 0.9, 1.3
    g = P.logpdf_and_gradient(fx(a, b), y)
    assert len(g["terms"]) == 12          # 4 per lower block pair (I >= J), mirrors folded
    d_a = d_b = 0.0
    for r in g["terms"]:
        k = int(r["row_scaled"]) + int(r["col_scaled"])       # coef = a^(2-k) b^k
        want = a ** (2 - k) * b ** k
        assert abs(r["coef"] - want) < 1e-14
        d_a += r["d_coef"] * (2 - k) * a ** (1 - k) * b ** k if k < 2 else 0.0
        d_b += r["d_coef"] * k * a ** (2 - k) * b ** (k - 1) if k > 0 else 0.0
    h = 1e-5
    fd_a = (P.logpdf(fx(a + h, b), y) - P.logpdf(fx(a - h, b), y)) / (2 * h)
    fd_b = (P.logpdf(fx(a, b + h), y) - P.logpdf(fx(a, b - h), y)) / (2 * h)
    assert abs(d_a - fd_a) <= 1e-6 * max(1.0, abs(fd_a)), (d_a, fd_a)
    assert abs(d_b - fd_b) <= 1e-6 * max(1.0, abs(fd_b)), (d_b, fd_b)


def test_logpdf_gradient_wrt_function_scales():
    """sigma(x) * f (product.jl:25-48): d logpdf / d sigma(x_i) per scale and input collection, (a) per term
    against a NumPy contraction of the oracle's G, (b) end to end -- two parametrised scales, one of them nested
    under the other and shared by two blocks -- against central differences of the GPU logpdf."""
    import np_terms
    rng = np.random.default_rng(33)
    x1, x2 = rng.standard_normal(120), rng.standard_normal(75)
    y = rng.standard_normal(195)

    def fx(th, ph):
        gpc = P.GPC()
        f1 = P.atomic(P.GP(P.Matern32Kernel()), gpc)
        f2 = P.atomic(P.GP(P.SEKernel()), gpc)
        s1 = lambda x: 1.0 + th * float(np.sum(np.sin(x)))
        s2 = lambda x: float(np.exp(ph * np.sum(x)))
        g1 = s1 * f1
        F = P.GPPP({"f1": f1, "g1": g1, "h": s2 * (g1 + f2)}, gpc)
        return F(P.BlockData([P.GPPPInput("h", x1), P.GPPPInput("g1", x2)]), 0.3), s1, s2

    th, ph = 0.4, 0.15
    f, s1, s2 = fx(th, ph)
    g = P.logpdf_and_gradient(f, y, scales=True)
    spec, grs = g["_spec"], g["_rowscale"]
    # (a) per term: 2 sum_j G_ij coef k_ij cs_j
    from oracle import stheno as ost  # noqa: F401  (oracle side of the same model)
    K = np_terms.dense_from_spec(spec)
    K = np.tril(K) + np.tril(K, -1).T
    C = K + 0.3 * np.eye(195)
    Ci = np.linalg.inv(C)
    al = Ci @ y
    G = 0.5 * (np.outer(al, al) - Ci)
    roff = np.concatenate([[0], np.cumsum(spec.row_len)])
    n_scaled = 0
    for t, (I, J, kind, ri, ci, coef, param, rs, cs) in enumerate(np_terms.spec_terms(spec)):
        if rs is None:
            assert grs[t] is None
            continue
        n_scaled += 1
        X, Y = spec.inputs[ri], spec.inputs[ci]
        d2 = ((X[:, :, None] - Y[:, None, :]) ** 2).sum(0)
        k = np_terms._kern(kind, d2, param)
        w = G[roff[I]:roff[I + 1], roff[J]:roff[J + 1]] * coef * k
        if cs is not None:
            w = w * cs[None, :]
        exp = 2.0 * w.sum(1)
        assert np.max(np.abs(grs[t] - exp)) <= 1e-9 * max(1.0, np.max(np.abs(exp))), t
    assert n_scaled >= 4
    # (b) chain rule onto the two parameters
    d_th = d_ph = 0.0
    seen = set()
    for r in g["scales"]:
        xs = np.asarray(r["x"].x if hasattr(r["x"], "x") else r["x"], dtype=np.float64)
        vals = r["values"]
        if np.allclose(vals, 1.0 + th * np.sin(xs)):            # s1 at this block's points
            d_th += float(r["d_values"] @ np.sin(xs))
            seen.add("s1")
        else:                                                   # s2 = exp(ph x): d/dph = x * values
            assert np.allclose(vals, np.exp(ph * xs))
            d_ph += float(r["d_values"] @ (xs * vals))
            seen.add("s2")
    assert seen == {"s1", "s2"} and len(g["scales"]) == 3       # s1 at x1 and at x2, s2 at x1
    h = 1e-5
    fd_th = (P.logpdf(fx(th + h, ph)[0], y) - P.logpdf(fx(th - h, ph)[0], y)) / (2 * h)
    fd_ph = (P.logpdf(fx(th, ph + h)[0], y) - P.logpdf(fx(th, ph - h)[0], y)) / (2 * h)
    assert abs(d_th - fd_th) <= 1e-6 * max(1.0, abs(fd_th)), (d_th, fd_th)
    assert abs(d_ph - fd_ph) <= 1e-6 * max(1.0, abs(fd_ph)), (d_ph, fd_ph)
    # the plain entry points are unchanged by the request
    g0 = P.logpdf_and_gradient(f, y)
    assert g0["logpdf"] == g["logpdf"] and np.array_equal(g0["_raw"][0], g["_raw"][0]) and g0["scales"] is None


def _kappa_prime(kind, d2):
    """d kappa / d (d^2) of the stationary kernels (independent restatement for the test)."""
    d = np.sqrt(d2)
    if kind == P.lib.SE:
        return -0.5 * np.exp(-0.5 * d2)
    if kind == P.lib.MATERN12:
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(d > 0, -0.5 * np.exp(-d) / d, 0.0)
    if kind == P.lib.MATERN32:
        return -1.5 * np.exp(-np.sqrt(3.0) * d)
    if kind == P.lib.MATERN52:
        return -(5.0 / 6.0) * (1.0 + np.sqrt(5.0) * d) * np.exp(-np.sqrt(5.0) * d)
    return np.zeros_like(d2)


def test_elbo_gradient_wrt_function_scales():
    """sigma(x) * f under the ELBO (round 3; product.jl:25-48 + sparse_finite_gp.jl:52-58): d elbo / d sigma(.) summed
    over K(z,z), K(x,z) (row side at x, column side at z) and diag K(x,x), against central differences of the elbo
    w.r.t. the parameters of two scales -- one nested under the other; observations in the scaled process, inducing
    points in two processes (one scaled, one not)."""
    rng = np.random.default_rng(44)
    x = rng.standard_normal(150)
    z1, z2 = np.linspace(-2, 2, 9), np.linspace(-1.5, 1.5, 7)
    y = rng.standard_normal(150)

    def build(th, ph):
        gpc = P.GPC()
        f1 = P.atomic(P.GP(P.Matern32Kernel()), gpc)
        f2 = P.atomic(P.GP(P.SEKernel()), gpc)
        s1 = lambda t: 1.0 + th * float(np.sum(np.sin(t)))
        s2 = lambda t: float(np.exp(ph * np.sum(t)))
        g1 = s1 * f1
        F = P.GPPP({"f1": f1, "f2": f2, "g1": g1, "h": s2 * (g1 + f2)}, gpc)
        fx = F(P.GPPPInput("h", x), 0.3)
        fz = F(P.BlockData([P.GPPPInput("g1", z1), P.GPPPInput("f2", z2)]), 1e-6)
        return P.VFE(fz), fx

    th, ph = 0.4, 0.15
    vfe, fx = build(th, ph)
    g = P.elbo_and_gradient(vfe, fx, y, scales=True)
    assert abs(g["elbo"] - P.elbo(vfe, fx, y)) <= 1e-10 * abs(g["elbo"])
    d_th = d_ph = 0.0
    for r in g["scales"]:
        xs = np.asarray(r["x"].x if hasattr(r["x"], "x") else r["x"], dtype=np.float64)
        vals = r["values"]
        if np.allclose(vals, 1.0 + th * np.sin(xs)):
            d_th += float(r["d_values"] @ np.sin(xs))
        else:
            assert np.allclose(vals, np.exp(ph * xs))
            d_ph += float(r["d_values"] @ (xs * vals))
    assert len(g["scales"]) >= 3            # s1 at x and at z1, s2 at x
    h = 1e-5
    fd_th = (P.elbo(*build(th + h, ph), y) - P.elbo(*build(th - h, ph), y)) / (2 * h)
    fd_ph = (P.elbo(*build(th, ph + h), y) - P.elbo(*build(th, ph - h), y)) / (2 * h)
    assert abs(d_th - fd_th) <= 1e-5 * max(1.0, abs(fd_th)), (d_th, fd_th)
    assert abs(d_ph - fd_ph) <= 1e-5 * max(1.0, abs(fd_ph)), (d_ph, fd_ph)


@pytest.mark.parametrize("recipe", [models.gppp_docstring, models.composite_kernels], ids=lambda r: r.__name__)
def test_logpdf_gradient_wrt_input_points(recipe):
    """sgp_logpdf_grad_x against a NumPy contraction of the oracle's G with the analytic kernel
    derivatives, per spec input (the transformed points the terms read)."""
    import np_terms
    rng = np.random.default_rng(21)
    Fo, Fp, fo, fp = both(recipe)
    names = list(fo)[:3]
    D = 2
    xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (150, 70, 131)][:len(names)]
    xo, xp = blockdata(names, xs, True)
    N = sum(x.shape[1] for x in xs)
    y = rng.standard_normal(N)
    _, _, G = oagp.logpdf_gradient_wrt_cov(Fo(xo, 0.3), y)
    g = P.logpdf_and_gradient(Fp(xp, 0.3), y, inputs=True)
    spec = g["_spec"]
    roff = np.concatenate([[0], np.cumsum(spec.row_len)])
    exp = [np.zeros_like(a) for a in spec.inputs]
    for (I, J, kind, ri, ci, coef, param, rs, cs) in np_terms.spec_terms(spec):
        X, Y = spec.inputs[ri], spec.inputs[ci]
        df = X[:, :, None] - Y[:, None, :]
        w = G[roff[I]:roff[I + 1], roff[J]:roff[J + 1]] * coef * _kappa_prime(kind, (df ** 2).sum(0))
        if rs is not None:
            w = w * rs[:, None]
        if cs is not None:
            w = w * cs[None, :]
        exp[ri] += 2.0 * 2.0 * (w[None, :, :] * df).sum(2)      # d(d2)/dx = 2 (x - x'), mirror block doubles
    assert len(g["inputs"]) == len(exp)
    for k, (a, e) in enumerate(zip(g["inputs"], exp)):
        assert a.shape == e.shape
        assert np.abs(a - e).max() <= 1e-8 * max(1.0, np.abs(e).max()), (k, np.abs(a - e).max())


def test_logpdf_input_gradient_matches_finite_differences():
    rng = np.random.default_rng(23)
    Xm = np.asfortranarray(rng.standard_normal((3, 300)))
    y = rng.standard_normal(300)
    v, l, s2 = 1.7, 0.8, 0.25
    f = np.sqrt(v) * P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 1.0 / l)
    g = P.logpdf_and_gradient(f(P.ColVecs(Xm), s2), y, inputs=True)
    (gx,) = g["inputs"]
    dX = gx / l                                   # the term reads X / l
    h = 1e-6
    for (d, i) in [(0, 0), (1, 17), (2, 299), (0, 150)]:
        Xp, Xn = Xm.copy(), Xm.copy()
        Xp[d, i] += h
        Xn[d, i] -= h
        fd = (P.logpdf(f(P.ColVecs(Xp), s2), y) - P.logpdf(f(P.ColVecs(Xn), s2), y)) / (2 * h)
        assert abs(dX[d, i] - fd) <= 1e-5 * max(1.0, abs(fd)), (d, i, dX[d, i], fd)


def test_elbo_and_sparse_posterior_row_chunked_path():
    """Very large N takes the row-chunked VFE pipeline (K(z,z) factored alone, K(x,z) solved and
    accumulated chunk by chunk); SGP_VFE_CHUNK forces it here: three chunks, the last partial and
    unaligned."""
    import os
    os.environ["SGP_VFE_CHUNK"] = "8192"
    try:
        _chunked_elbo_body()
    finally:
        del os.environ["SGP_VFE_CHUNK"]


def _chunked_elbo_body():
    rng = np.random.default_rng(41)
    N, M, D = 20000 + 37, 150, 2
    X = np.asfortranarray(rng.standard_normal((D, N)))
    Z = np.asfortranarray(rng.standard_normal((D, M)))
    y = rng.standard_normal(N)
    fo = ost.atomic(oagp.GP(0.2, okf.Matern52Kernel()), ost.GPC())
    fp = P.atomic(P.GP(0.2, P.Matern52Kernel()), P.GPC())
    noise = 0.1 + rng.random(N)
    eo = oagp.elbo(oagp.VFE(fo(okf.ColVecs(Z), 1e-6)), fo(okf.ColVecs(X), noise), y)
    ep = P.elbo(P.VFE(fp(P.ColVecs(Z), 1e-6)), fp(P.ColVecs(X), noise), y)
    assert abs(ep - eo) <= 1e-9 * abs(eo), (ep, eo)
    Xs = np.asfortranarray(rng.standard_normal((D, 50)))
    po = oagp.posterior_vfe(oagp.VFE(fo(okf.ColVecs(Z), 1e-6)), fo(okf.ColVecs(X), noise), y)
    pp = P.posterior(P.VFE(fp(P.ColVecs(Z), 1e-6)), fp(P.ColVecs(X), noise), y)
    mo, vo = po.mean_and_var(okf.ColVecs(Xs))
    mp_, vp = P.mean_and_var(pp(P.ColVecs(Xs), 0.0))
    assert rel(mp_, mo) < 1e-7 and np.abs(vp - vo).max() < 1e-8


# ---- reverse-mode gradient of the elbo --------------------------------------
@pytest.mark.parametrize("recipe", [models.gppp_docstring, models.composite_kernels], ids=lambda r: r.__name__)
def test_elbo_gradient_against_oracle_cotangents(recipe):
    """sgp_elbo_grad against oracle.abstractgps.elbo_gradient_wrt_cov (itself checked against finite
    differences on the CPU): y / noise / Sigma_z cotangents directly, the per-term numbers by
    contracting the oracle's dKzz, dKxz, dvar with the NumPy term kernels."""
    import np_terms
    rng = np.random.default_rng(11)
    Fo, Fp, fo, fp = both(recipe)
    names = list(fo)[:3]
    D = 2
    xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (260, 90, 131)][:len(names)]
    zs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (40, 33, 70)][:len(names)]
    xo, xp = blockdata(names, xs, True)
    zo, zp = blockdata(names, zs, True)
    N, M = sum(x.shape[1] for x in xs), sum(z.shape[1] for z in zs)
    y = rng.standard_normal(N)
    for noise in (0.3, 0.1 + rng.random(N)):
        go = oagp.elbo_gradient_wrt_cov(oagp.VFE(Fo(zo, 1e-6)), Fo(xo, noise), y)
        g = P.elbo_and_gradient(P.VFE(Fp(zp, 1e-6)), Fp(xp, noise), y)
        assert abs(g["elbo"] - go["elbo"]) <= 1e-9 * abs(go["elbo"])
        assert abs(g["elbo"] - P.elbo(P.VFE(Fp(zp, 1e-6)), Fp(xp, noise), y)) <= 1e-12 * abs(g["elbo"])
        assert rel(g["y"], go["y"]) < 1e-7 and rel(g["mean"], go["mean"]) < 1e-7
        assert rel(g["noise"], go["noise"]) < 1e-7
        assert rel(g["var"], go["var"]) < 1e-12
        # Sigma_z = 1e-6 I makes Kzz ill-conditioned (cond ~ 1e6): cotangents carry ~1e-6 relative noise
        scale = np.abs(go["Kzz"]).max()
        assert abs(g["z_noise"] - np.trace(go["Kzz"])) <= 1e-5 * scale * M
        specs, raw = g["_specs"], g["_raw"]
        for key, G in (("zz", go["Kzz"]), ("xz", go["Kxz"])):
            exp = _oracle_term_grads(specs[key], G)
            gc, gs = raw[key]
            ref = max(1.0, max(abs(e[0]) for e in exp), max(abs(e[1]) for e in exp))
            for t, (ec, es) in enumerate(exp):
                assert abs(gc[t] - ec) <= 2e-5 * ref, (key, t, gc[t], ec)
                assert abs(gs[t] - es) <= 2e-5 * ref, (key, t, gs[t], es)
        # diagonal terms: sum_i dvar_i rs_i cs_i k(x_i, x_i)
        gcd, gsd = raw["xx"]
        roff = np.concatenate([[0], np.cumsum(specs["xx"].row_len)])
        for t, (I, J, kind, ri, ci, coef, param, rs, cs) in enumerate(np_terms.spec_terms(specs["xx"])):
            if I != J:
                assert gcd[t] == 0.0 and gsd[t] == 0.0
                continue
            X, Y = specs["xx"].inputs[ri], specs["xx"].inputs[ci]
            k = np_terms._kern(kind, ((X - Y) ** 2).sum(0), param)
            w = go["var"][roff[I]:roff[I + 1]] * (1.0 if rs is None else rs) * (1.0 if cs is None else cs)
            assert abs(gcd[t] - (w * k).sum()) <= 1e-10 * max(1.0, abs((w * k).sum()))


def test_elbo_gradient_with_dense_inducing_noise():
    """Sigma_z a full positive definite matrix (the reference treats dense / diagonal / isotropic noise alike:
    test/affine_transformations/test_util.jl:114-134): the bound, its data cotangents, and d elbo / d Sigma_z = the whole
    M x M cotangent of Kzz + Sigma_z, against the oracle's."""
    rng = np.random.default_rng(23)
    Fo, Fp, fo, fp = both(models.gppp_docstring)
    names = list(fo)[:2]
    D = 2
    xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (230, 141)][:len(names)]
    zs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (37, 30)][:len(names)]
    xo, xp = blockdata(names, xs, True)
    zo, zp = blockdata(names, zs, True)
    N, M = sum(x.shape[1] for x in xs), sum(z.shape[1] for z in zs)
    y = rng.standard_normal(N)
    Q = rng.standard_normal((M, M))
    Sz = 1e-3 * np.eye(M) + 1e-3 * (Q @ Q.T) / M
    noise = 0.1 + rng.random(N)
    go = oagp.elbo_gradient_wrt_cov(oagp.VFE(Fo(zo, Sz)), Fo(xo, noise), y)
    g = P.elbo_and_gradient(P.VFE(Fp(zp, Sz)), Fp(xp, noise), y)
    assert abs(g["elbo"] - go["elbo"]) <= 1e-9 * abs(go["elbo"])
    assert abs(g["elbo"] - P.elbo(P.VFE(Fp(zp, Sz)), Fp(xp, noise), y)) <= 1e-12 * abs(g["elbo"])
    assert rel(g["y"], go["y"]) < 1e-7 and rel(g["noise"], go["noise"]) < 1e-7
    assert g["z_noise"].shape == (M, M)
    assert np.abs(g["z_noise"] - go["Kzz"]).max() <= 1e-6 * np.abs(go["Kzz"]).max()
    assert np.abs(g["z_noise"] - g["z_noise"].T).max() <= 1e-9 * np.abs(go["Kzz"]).max()


def test_elbo_gradient_matches_finite_differences_of_hyperparameters():
    """End to end: variance, lengthscale and noise of s * stretch(GP(Matern52), 1/l) through the
    VFE bound, against central differences of the GPU elbo itself."""
    rng = np.random.default_rng(13)
    X = P.ColVecs(rng.standard_normal((2, 700)))
    Z = P.ColVecs(rng.standard_normal((2, 60)))
    y = rng.standard_normal(700)

    def model(v, l):
        return np.sqrt(v) * P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 1.0 / l)

    def bound(v, l, s2):
        f = model(v, l)
        return P.elbo(P.VFE(f(Z, 1e-4)), f(X, s2), y)

    v, l, s2 = 1.7, 0.8, 0.25
    f = model(v, l)
    g = P.elbo_and_gradient(P.VFE(f(Z, 1e-4)), f(X, s2), y)
    allt = g["zz_terms"] + g["xz_terms"] + g["xx_terms"]
    assert len(g["zz_terms"]) == 1 and len(g["xz_terms"]) == 1 and len(g["xx_terms"]) == 1
    d_v = sum(t["d_coef"] for t in allt)
    d_l = -(1.0 / l) * sum(t["d_inscale"] for t in allt)
    h = 1e-5
    fd_v = (bound(v + h, l, s2) - bound(v - h, l, s2)) / (2 * h)
    fd_l = (bound(v, l + h, s2) - bound(v, l - h, s2)) / (2 * h)
    fd_s = (bound(v, l, s2 + h) - bound(v, l, s2 - h)) / (2 * h)
    assert abs(d_v - fd_v) <= 1e-5 * max(1.0, abs(fd_v)), (d_v, fd_v)
    assert abs(d_l - fd_l) <= 1e-5 * max(1.0, abs(fd_l)), (d_l, fd_l)
    assert abs(g["noise"] - fd_s) <= 1e-5 * max(1.0, abs(fd_s)), (g["noise"], fd_s)


def test_elbo_input_gradients_match_finite_differences():
    """d elbo / d (data points) and d elbo / d (inducing points): sgp_elbo_grad_x against central
    differences of the GPU elbo (Matern-5/2, stretched inputs: the terms read X / l and Z / l)."""
    rng = np.random.default_rng(29)
    Xm = np.asfortranarray(rng.standard_normal((2, 500)))
    Zm = np.asfortranarray(rng.standard_normal((2, 40)))
    y = rng.standard_normal(500)
    v, l, s2 = 1.7, 0.8, 0.25
    f = np.sqrt(v) * P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 1.0 / l)

    def bound(Xa, Za):
        return P.elbo(P.VFE(f(P.ColVecs(Za), 1e-4)), f(P.ColVecs(Xa), s2), y)

    g = P.elbo_and_gradient(P.VFE(f(P.ColVecs(Zm), 1e-4)), f(P.ColVecs(Xm), s2), y, inputs=True)
    specs = g["_specs"]
    (gzz,) = g["zz_inputs"]
    assert len(g["xz_inputs"]) == 2
    (t,) = g["xz_terms"]
    gx_x, gx_z = g["xz_inputs"][t["row_input"]], g["xz_inputs"][t["col_input"]]
    assert gx_x.shape == Xm.shape and gx_z.shape == Zm.shape and gzz.shape == Zm.shape
    dX, dZ = gx_x / l, (gx_z + gzz) / l
    h = 1e-6
    for (d, i) in [(0, 0), (1, 250), (0, 499)]:
        Xp, Xn = Xm.copy(), Xm.copy()
        Xp[d, i] += h
        Xn[d, i] -= h
        fd = (bound(Xp, Zm) - bound(Xn, Zm)) / (2 * h)
        assert abs(dX[d, i] - fd) <= 2e-5 * max(1.0, abs(fd)), ("x", d, i, dX[d, i], fd)
    for (d, j) in [(0, 0), (1, 20), (0, 39)]:
        Zp, Zn = Zm.copy(), Zm.copy()
        Zp[d, j] += h
        Zn[d, j] -= h
        fd = (bound(Xm, Zp) - bound(Xm, Zn)) / (2 * h)
        assert abs(dZ[d, j] - fd) <= 2e-5 * max(1.0, abs(fd)), ("z", d, j, dZ[d, j], fd)


@pytest.mark.parametrize("N,M", [(1, 1), (5, 2), (128, 128), (129, 3)])
def test_gradients_at_tiny_and_tile_boundary_sizes(N, M):
    """logpdf / elbo gradients where the 128-padding dominates: against finite differences."""
    rng = np.random.default_rng(100 + N)
    Xm = np.asfortranarray(rng.standard_normal((2, N)))
    Zm = np.asfortranarray(rng.standard_normal((2, M)))
    y = rng.standard_normal(N)
    s2, h = 0.3, 1e-6

    def model(v):
        return np.sqrt(v) * P.atomic(P.GP(P.SEKernel()), P.GPC())

    g = P.logpdf_and_gradient(model(1.3)(P.ColVecs(Xm), s2), y, inputs=True)
    fd_v = (P.logpdf(model(1.3 + h)(P.ColVecs(Xm), s2), y) - P.logpdf(model(1.3 - h)(P.ColVecs(Xm), s2), y)) / (2 * h)
    fd_s = (P.logpdf(model(1.3)(P.ColVecs(Xm), s2 + h), y) - P.logpdf(model(1.3)(P.ColVecs(Xm), s2 - h), y)) / (2 * h)
    assert abs(g["terms"][0]["d_coef"] - fd_v) <= 1e-6 * max(1.0, abs(fd_v))
    assert abs(g["noise"] - fd_s) <= 1e-6 * max(1.0, abs(fd_s))
    Xp, Xn = Xm.copy(), Xm.copy()
    Xp[1, N - 1] += h
    Xn[1, N - 1] -= h
    fd_x = (P.logpdf(model(1.3)(P.ColVecs(Xp), s2), y) - P.logpdf(model(1.3)(P.ColVecs(Xn), s2), y)) / (2 * h)
    assert abs(g["inputs"][0][1, N - 1] - fd_x) <= 1e-6 * max(1.0, abs(fd_x))

    def bound(v, s):
        f = model(v)
        return P.elbo(P.VFE(f(P.ColVecs(Zm), 1e-3)), f(P.ColVecs(Xm), s), y)

    f = model(1.3)
    ge = P.elbo_and_gradient(P.VFE(f(P.ColVecs(Zm), 1e-3)), f(P.ColVecs(Xm), s2), y)
    assert abs(ge["elbo"] - bound(1.3, s2)) <= 1e-12 * max(1.0, abs(ge["elbo"]))
    d_v = sum(t["d_coef"] for t in ge["zz_terms"] + ge["xz_terms"] + ge["xx_terms"])
    fd_v = (bound(1.3 + h, s2) - bound(1.3 - h, s2)) / (2 * h)
    fd_s = (bound(1.3, s2 + h) - bound(1.3, s2 - h)) / (2 * h)
    assert abs(d_v - fd_v) <= 1e-5 * max(1.0, abs(fd_v)), (d_v, fd_v)
    assert abs(ge["noise"] - fd_s) <= 1e-5 * max(1.0, abs(fd_s)), (ge["noise"], fd_s)


def test_input_gradients_chain_through_model_transformations():
    """g["x"] / g["z"]: gradients w.r.t. the user's own inputs (host chain rule through kernel
    length-scales, stretch, periodic, select over the device gradient), against central
    differences of the GPU logpdf / elbo."""
    rng = np.random.default_rng(37)
    gpc = P.GPC()
    a = P.atomic(P.GP(P.with_lengthscale(P.Matern52Kernel(), 0.7)), gpc)
    b = P.atomic(P.GP(P.SEKernel()), gpc)
    F = P.GPPP({"f1": P.stretch(a, np.array([0.5, 2.0])), "f2": P.periodic(b, 0.6),
                "f3": P.stretch(a, 1.3) + P.select(P.stretch(a, 0.8), [1, 0])}, gpc)
    mats = [rng.standard_normal((2, 60)), rng.standard_normal((1, 45)), rng.standard_normal((2, 50))]
    names = ["f1", "f2", "f3"]

    def data(ms):
        return P.BlockData([P.GPPPInput(k, P.ColVecs(m) if m.shape[0] == 2 else m.reshape(-1))
                            for k, m in zip(names, ms)])

    N = sum(m.shape[1] for m in mats)
    y = rng.standard_normal(N)
    g = P.logpdf_and_gradient(F(data(mats), 0.2), y, inputs=True)
    h = 1e-6
    for I, (d, i) in [(0, (1, 7)), (1, (0, 30)), (2, (0, 49)), (2, (1, 0))]:
        mp_, mn_ = [q.copy() for q in mats], [q.copy() for q in mats]
        mp_[I][d, i] += h
        mn_[I][d, i] -= h
        fd = (P.logpdf(F(data(mp_), 0.2), y) - P.logpdf(F(data(mn_), 0.2), y)) / (2 * h)
        assert abs(g["x"][I][d, i] - fd) <= 2e-5 * max(1.0, abs(fd)), (I, d, i, g["x"][I][d, i], fd)
    # elbo: inducing points on f1 and f3
    zm = [rng.standard_normal((2, 12)), rng.standard_normal((2, 9))]

    def zdata(zs):
        return P.BlockData([P.GPPPInput("f1", P.ColVecs(zs[0])), P.GPPPInput("f3", P.ColVecs(zs[1]))])

    def bound(ms, zs):
        return P.elbo(P.VFE(F(zdata(zs), 1e-4)), F(data(ms), 0.2), y)

    ge = P.elbo_and_gradient(P.VFE(F(zdata(zm), 1e-4)), F(data(mats), 0.2), y, inputs=True)
    for J, (d, j) in [(0, (0, 3)), (1, (1, 8))]:
        zp, zn = [q.copy() for q in zm], [q.copy() for q in zm]
        zp[J][d, j] += h
        zn[J][d, j] -= h
        fd = (bound(mats, zp) - bound(mats, zn)) / (2 * h)
        assert abs(ge["z"][J][d, j] - fd) <= 5e-5 * max(1.0, abs(fd)), ("z", J, d, j, ge["z"][J][d, j], fd)
    for I, (d, i) in [(0, (0, 0)), (1, (0, 10)), (2, (1, 25))]:
        mp_, mn_ = [q.copy() for q in mats], [q.copy() for q in mats]
        mp_[I][d, i] += h
        mn_[I][d, i] -= h
        fd = (bound(mp_, zm) - bound(mn_, zm)) / (2 * h)
        assert abs(ge["x"][I][d, i] - fd) <= 5e-5 * max(1.0, abs(fd)), ("x", I, d, i, ge["x"][I][d, i], fd)


# ---- ill-conditioned covariances: the panel solves must be as accurate as LAPACK's ---------------
def _illcond_cases():
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "illcond_truth.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _illcond_cases(), ids=lambda c: f"N{c['N']}-noise{c['noise']:g}")
def test_logpdf_ill_conditioned_against_60_digit_reference(case):
    """cond(C) = 1e6 .. 1e12 (SE kernel, sorted 1-D inputs, tiny noise).  A product with an explicit
    inverse in the panel solve loses 2+ digits here and reports a spurious PosDefException at
    noise 1e-12; the refined solves must stay within a small multiple of LAPACK's own error
    (the oracle) against the 60-digit value, and within the north star's 1e-8 of the oracle
    wherever the oracle itself is that accurate."""
    x, y, s2 = np.array(case["x"]), np.array(case["y"]), case["noise"]
    truth = float(case["logpdf"])
    lo = oagp.logpdf(ost.atomic(oagp.GP(okf.SEKernel()), ost.GPC())(x, s2), y)
    lp = P.logpdf(P.atomic(P.GP(P.SEKernel()), P.GPC())(x, s2), y)   # must not raise PosDef
    err_o, err_p = abs(lo - truth) / abs(truth), abs(lp - truth) / abs(truth)
    assert err_p <= 10.0 * max(err_o, 1e-13), (err_p, err_o)
    if err_o < 1e-9:
        assert abs(lp - lo) <= 1e-8 * abs(lo)
