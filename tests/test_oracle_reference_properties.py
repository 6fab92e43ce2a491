"""The reference's own property tests for the hot path, ported onto the CPU oracle (no GPU).

These are the only pins the reference holds at this boundary: it has
no numeric golden vectors.  Each test cites the reference test it restates."""
import numpy as np
import pytest

import oracle.abstractgps as agp
import oracle.kernelfunctions as kf
import oracle.stheno as st


def rng():
    return np.random.default_rng(123456)


def abstractgp_interface_tests(f, f2, x0, x1, x2, x3, atol=1e-9, rtol=1e-9):
    """test/test_util.jl:113-168.  x0, x1 are valid inputs for f; x2, x3 for f2;
    len(x0) != len(x1), len(x0) != len(x2), len(x0) == len(x3)."""
    n0 = len(x0)
    assert f.mean(x0).shape == (n0,)
    assert n0 != len(x1) and n0 != len(x2) and n0 == len(x3)
    K02 = st.cov4(f, f2, x0, x2)
    assert K02.shape == (n0, len(x2))
    np.testing.assert_allclose(K02, st.cov4(f2, f, x2, x0).T, rtol=rtol, atol=atol)
    K = f.cov(x0)
    assert K.shape == (n0, n0)
    np.testing.assert_allclose(K, st.cov4(f, f, x0, x0), rtol=rtol, atol=atol)
    assert np.linalg.eigvalsh(K).min() > -atol
    np.testing.assert_allclose(K, K.T, rtol=rtol, atol=atol)
    K01 = f.cov(x0, x1)
    assert K01.shape == (n0, len(x1))
    np.testing.assert_allclose(K01, st.cov4(f, f, x0, x1), rtol=rtol, atol=atol)
    d = st.var4(f, f2, x0, x3)
    assert d.shape == (n0,)
    np.testing.assert_allclose(d, np.diag(st.cov4(f, f2, x0, x3)), rtol=rtol, atol=atol)
    np.testing.assert_allclose(d, st.var4(f2, f, x3, x0), rtol=rtol, atol=atol)
    np.testing.assert_allclose(f.var(x0, x0), np.diag(f.cov(x0, x0)), rtol=rtol, atol=atol)
    np.testing.assert_allclose(f.var(x0), np.diag(K), rtol=rtol, atol=atol)


def test_atomic_gp():
    """test/gp/atomic_gp.jl:6-38."""
    gpc = st.GPC()
    k = kf.SqExponentialKernel()
    f = st.atomic(agp.GP(np.sin, k), gpc)
    x, x2 = np.linspace(-1, 1, 5), np.linspace(-1, 1, 6)
    assert np.array_equal(f.mean(x), np.sin(x))
    assert np.array_equal(f.cov(x), kf.kernelmatrix(k, x))
    r = rng()
    x, x2 = r.standard_normal(5), r.standard_normal(6)
    gpc = st.GPC()
    f1 = st.atomic(agp.GP(agp.ZeroMean(), kf.SEKernel()), gpc)
    f2 = st.atomic(agp.GP(agp.ConstMean(5), kf.SEKernel()), gpc)
    assert np.array_equal(f1.mean(x), np.zeros(5)) and np.array_equal(f2.mean(x), np.full(5, 5.0))
    assert np.array_equal(st.cov4(f1, f2, x, x2), np.zeros((5, 6)))
    assert np.array_equal(f1.var(x), np.ones(5))
    np.testing.assert_allclose(st.cov4(f1, f1, x2, x), st.cov4(f1, f1, x, x2).T)
    abstractgp_interface_tests(f1, f2, x, x2, x2, x)


def test_cross_correctness():
    """test/affine_transformations/cross.jl:32-77."""
    gpc = st.GPC()
    f1 = st.atomic(agp.GP(np.sin, kf.SEKernel()), gpc)
    f2 = st.atomic(agp.GP(np.cos, kf.SEKernel()), gpc)
    f3, f4, f5 = st.cross([f1, f2]), st.cross([f1]), st.cross([f2])
    x1, x2 = np.linspace(-5, 5, 2), np.linspace(-5, 5, 3)
    x3, x4, x5 = st.BlockData([x1, x2]), st.BlockData([x1]), st.BlockData([x2])
    assert np.array_equal(agp.mean(f3(x3)), np.concatenate([agp.mean(f1(x1)), agp.mean(f2(x2))]))
    C = np.block([[f1.cov(x1), st.cov4(f1, f2, x1, x2)], [st.cov4(f2, f1, x2, x1), f2.cov(x2)]])
    np.testing.assert_allclose(f3.cov(x3), C)
    assert np.array_equal(agp.cov(f3(x3), f3(x3)), f3.cov(x3))
    assert np.array_equal(f4.cov(x4), f1.cov(x1)) and np.array_equal(f5.cov(x5), f2.cov(x2))
    assert np.array_equal(agp.cov(f1(x1), f2(x2)), agp.cov(f4(x4), f5(x5)))
    assert np.array_equal(agp.cov(f3(x3), f4(x4)), np.vstack([f1.cov(x1), st.cov4(f2, f1, x2, x1)]))
    assert np.array_equal(agp.cov(f5(x5), f3(x3)), np.hstack([st.cov4(f2, f1, x2, x1), f2.cov(x2)]))
    assert np.array_equal(agp.cov(f3(x3), f4(x4)), agp.cov(f3(x3), f1(x1)))
    r = rng()
    x0 = st.BlockData([np.linspace(-1, 1, 3), np.linspace(2, 4, 3)])
    xb = st.BlockData([r.standard_normal(5), r.standard_normal(5)])
    abstractgp_interface_tests(f3, f1, x0, xb, r.standard_normal(3), r.standard_normal(6))


def test_addition():
    """test/affine_transformations/addition.jl:2-60."""
    r = rng()
    gpc = st.GPC()
    X, X2 = kf.ColVecs(r.standard_normal((2, 5))), kf.ColVecs(r.standard_normal((2, 6)))
    f1 = st.atomic(agp.GP(1, kf.SEKernel()), gpc)
    f2 = st.atomic(agp.GP(2, kf.SEKernel()), gpc)
    f3 = f1 + f2
    f4 = f1 + f3
    f5 = f3 + f4
    for fp, fa, fb in [(f3, f1, f2), (f4, f1, f3), (f5, f3, f4)]:
        S = fa.cov(X) + fb.cov(X) + st.cov4(fa, fb, X, X) + st.cov4(fb, fa, X, X)
        np.testing.assert_allclose(fp.mean(X), fa.mean(X) + fb.mean(X))
        np.testing.assert_allclose(fp.cov(X), S)
        np.testing.assert_allclose(st.cov4(fp, fa, X, X2), st.cov4(fa, fa, X, X2) + st.cov4(fb, fa, X, X2))
        np.testing.assert_allclose(st.cov4(fa, fp, X, X2), st.cov4(fa, fb, X, X2) + st.cov4(fa, fa, X, X2))
        np.testing.assert_allclose(st.cov4(fp, fp, X2, X), st.cov4(fp, fp, X, X2).T)
    x0, x1, x2, x3 = r.standard_normal(4), r.standard_normal(3), r.standard_normal(3), r.standard_normal(4)
    abstractgp_interface_tests(f3, f1, x0, x1, x2, x3)
    abstractgp_interface_tests(f2 - f1, f1, x0, x1, x2, x3)
    c, f = float(r.standard_normal()), st.atomic(agp.GP(5, kf.SEKernel()), st.GPC())
    Xc = kf.ColVecs(r.standard_normal((6, 5)))
    assert np.array_equal((f + c).mean(Xc), f.mean(Xc) + c) and np.array_equal((c + f).cov(Xc), f.cov(Xc))
    assert np.array_equal((f - c).mean(Xc), f.mean(Xc) - c) and np.array_equal((c - f).mean(Xc), c - f.mean(Xc))
    assert np.array_equal((c - f).cov(Xc), f.cov(Xc))
    x = r.standard_normal(11)
    assert np.array_equal((f + np.sin).mean(x), f.mean(x) + np.sin(x)) and np.array_equal((np.sin + f).cov(x), f.cov(x))


def test_product():
    """test/affine_transformations/product.jl:8-111."""
    r = rng()
    X, X2 = kf.ColVecs(r.standard_normal((2, 3))), kf.ColVecs(r.standard_normal((2, 5)))
    g1, c, c2 = st.atomic(agp.GP(1, kf.SEKernel()), st.GPC()), -4.3, 2.1
    g2, g2p = c * g1, g1 * c2
    g3, g3p = c * g2, g2p * c2
    g4, g4p = c * g3, g3p * c2
    assert np.array_equal(g2.mean(X), c * g1.mean(X)) and np.array_equal(g4p.mean(X), g3p.mean(X) * c2)
    np.testing.assert_allclose(g2.cov(X), c ** 2 * g1.cov(X))
    np.testing.assert_allclose(g4.cov(X), c ** 2 * g3.cov(X))
    K = st.cov4(g1, g1, X, X2)
    np.testing.assert_allclose(st.cov4(g4, g1, X, X2), c ** 3 * K)
    np.testing.assert_allclose(st.cov4(g3, g3p, X, X2), (c * c2) ** 2 * K)
    np.testing.assert_allclose(st.cov4(g2, g4p, X, X2), (c * c2 ** 3) * K)
    with pytest.raises(ValueError):
        g1 * g2
    fs, fc = (lambda x: float(np.sum(np.sin(x)))), (lambda x: float(np.sum(np.cos(x))))
    h2, h2p = fs * g1, g1 * fc
    h3, h3p = fs * h2, h2p * fc
    fX = np.array([fs(X.X[:, i]) for i in range(3)])
    fX2p = np.array([fc(X2.X[:, i]) for i in range(5)])
    np.testing.assert_allclose(h2.cov(X), fX[:, None] * g1.cov(X) * fX[None, :])
    np.testing.assert_allclose(st.cov4(h3, g1, X, X2), (fX ** 2)[:, None] * K)
    np.testing.assert_allclose(st.cov4(h2, h3p, X, X2), fX[:, None] * K * (fX2p ** 2)[None, :])
    assert np.array_equal(st.cov4(g1, h2, X2, X), st.cov4(h2, g1, X, X2).T)
    x0, x1 = np.linspace(-1, 1, 3), np.linspace(-0.5, 1.5, 5)
    f1 = st.atomic(agp.GP(np.cos, kf.SEKernel()), st.GPC())
    abstractgp_interface_tests(5 * f1, f1, x0, x1, r.standard_normal(5), r.standard_normal(3))
    abstractgp_interface_tests(np.sin * f1, f1, x0, x1, r.standard_normal(5), r.standard_normal(3))


def test_compose_and_warps():
    """test/affine_transformations/compose.jl:2-170."""
    r = rng()
    gpc = st.GPC()
    x, x2 = r.standard_normal(5), r.standard_normal(3)
    f = st.atomic(agp.GP(np.sin, kf.SEKernel()), gpc)
    h = st.atomic(agp.GP(np.exp, kf.ExponentialKernel()), gpc)
    fg = st.compose(f, np.cos)
    assert np.array_equal(fg.mean(x), f.mean(np.cos(x))) and np.array_equal(fg.cov(x), f.cov(np.cos(x)))
    assert np.array_equal(st.cov4(fg, fg, x, x2), f.cov(np.cos(x), np.cos(x2)))
    assert np.array_equal(st.cov4(fg, f, x, x2), f.cov(np.cos(x), x2))
    assert np.array_equal(st.cov4(fg, h, x, x2), np.zeros((5, 3))) and np.array_equal(st.cov4(h, fg, x, x2), np.zeros((5, 3)))
    abstractgp_interface_tests(fg, f, r.standard_normal(4), r.standard_normal(3), r.standard_normal(3), r.standard_normal(4))
    abstractgp_interface_tests(st.stretch(f, 0.1), f, r.standard_normal(4), r.standard_normal(3), r.standard_normal(3), r.standard_normal(4))
    lam = 0.51
    f0 = st.atomic(agp.GP(1.3, kf.SEKernel()), st.GPC())
    g = st.stretch(f0, lam)
    xs = r.standard_normal(1)
    assert st.cov4(f0, g, np.zeros(1), np.zeros(1))[0, 0] == 1.0
    assert st.cov4(f0, g, lam * xs, xs)[0, 0] == 1.0
    Xc = r.standard_normal((11, 1))
    assert st.cov4(f0, g, kf.ColVecs(lam * Xc), kf.ColVecs(Xc))[0, 0] == 1.0
    lv = r.standard_normal(7)
    gv = st.stretch(f0, lv)
    X7 = r.standard_normal((7, 1))
    # The reference asserts `== 1.0` here; with the GEMM-trick distances that bit-exactness hinges
    # on BLAS-vs-sum summation order (|a|^2 + |a|^2 - 2 a'a), which NumPy does not reproduce, so
    # the oracle is held to 4 ulp.  The HIP path sums (a_d - b_d)^2 directly and IS exactly 1.0
    # (tests/test_gpu_parity.py::test_exact_identities_from_reference_tests).
    assert abs(st.cov4(f0, gv, kf.ColVecs(np.diag(lv) @ X7), kf.ColVecs(X7))[0, 0] - 1.0) < 1e-15
    A = r.standard_normal((7, 7))
    assert abs(st.cov4(f0, st.stretch(f0, A), kf.ColVecs(A @ X7), kf.ColVecs(X7))[0, 0] - 1.0) < 1e-15
    D, N = 6, 3
    X = r.standard_normal((D, N))
    for idx, Xf in [(0, X[0, :]), ([0, 2], kf.ColVecs(X[[0, 2], :]))]:
        gs = st.select(f0, idx)
        np.testing.assert_allclose(st.cov4(f0, gs, Xf, kf.ColVecs(X)), f0.cov(Xf, Xf))
        np.testing.assert_allclose(st.cov4(f0, gs, Xf, kf.ColVecs(X)), gs.cov(kf.ColVecs(X), kf.ColVecs(X)))
    a = float(r.standard_normal())
    gsh = st.shift(f0, a)
    xv = r.standard_normal(N)
    np.testing.assert_allclose(st.cov4(f0, gsh, xv - a, xv), f0.cov(xv - a, xv - a))
    av = r.standard_normal(D)
    gshv = st.shift(f0, av)
    np.testing.assert_allclose(st.cov4(f0, gshv, kf.ColVecs(X - av[:, None]), kf.ColVecs(X)), gshv.cov(kf.ColVecs(X)))
    fp_ = st.atomic(agp.GP(kf.SEKernel()), st.GPC())
    gp_ = st.periodic(fp_, 2.0)
    np.testing.assert_allclose(agp.cov(gp_(np.array([0.0])), gp_(np.array([1.0]))),
                               agp.cov(gp_(np.array([0.0])), gp_(np.array([3.0]))))


def test_gppp_external_consistency_and_split():
    """test/gaussian_process_probabilistic_programme.jl:3-42."""
    r = rng()
    x = st.BlockData(r.standard_normal(5), r.standard_normal(4))
    a, b = st.split(x, r.standard_normal(9))
    assert len(a) == 5 and len(b) == 4
    A, B = st.split(x, r.standard_normal((9, 3)))
    assert A.shape == (5, 3) and B.shape == (4, 3)
    gpc = st.GPC()
    f1 = st.atomic(agp.GP(np.sin, kf.SEKernel()), gpc)
    f2 = st.atomic(agp.GP(np.cos, kf.Matern52Kernel()), gpc)
    f3 = f1 + 3 * f2
    f = st.GPPP({"f1": f1, "f2": f2, "f3": f3}, gpc)
    x0, x1 = st.GPPPInput("f1", r.standard_normal(4)), st.GPPPInput("f3", r.standard_normal(3))
    assert np.array_equal(f1.mean(x0.x), f.mean(x0)) and np.array_equal(f3.mean(x1.x), f.mean(x1))
    assert np.array_equal(f1.cov(x0.x), f.cov(x0)) and np.array_equal(f3.cov(x1.x), f.cov(x1))
    assert np.array_equal(st.cov4(f1, f3, x0.x, x1.x), f.cov(x0, x1))
    assert np.array_equal(st.var4(f3, f1, x1.x, x1.x), f3.var(x1.x) * 0 + st.var4(f3, f1, x1.x, x1.x))
    y = agp.rand(f(x1), r.standard_normal(3))
    c1 = agp.posterior(f3(x1.x), y).cov(x1.x)
    c2 = agp.posterior(f(x1), y).cov(x1)
    assert np.array_equal(c1, c2)
    # internal consistency over the input-type permutations of :47-86
    xb = st.BlockData([st.GPPPInput("f2", r.standard_normal(3)), st.GPPPInput("f3", r.standard_normal(2))])
    for u, v in [(x0, x1), (x0, xb), (xb, x0)]:
        K = f.cov(u)
        np.testing.assert_allclose(K, K.T, atol=1e-12)
        assert np.linalg.eigvalsh(K).min() > -1e-9
        np.testing.assert_allclose(f.var(u), np.diag(K), atol=1e-12)
        np.testing.assert_allclose(f.cov(u, v), f.cov(v, u).T, atol=1e-12)
        yy = agp.rand(f(u, 0.1), r.standard_normal(len(u)))
        assert np.isfinite(agp.logpdf(f(u, 0.1), yy))
        post = agp.posterior(f(u, 0.1), yy)
        assert post.mean(v).shape == (len(v),) and np.all(post.var(v) > -1e-9)


def test_finite_gp_statistics_and_rand():
    """test/gp/util.jl:9-47."""
    r = rng()
    x, x2 = r.standard_normal(1), r.standard_normal(9)
    f = st.atomic(agp.GP(np.sin, kf.SEKernel()), st.GPC())
    fx, fx2 = agp.FiniteGP(f, x, np.zeros((1, 1))), agp.FiniteGP(f, x2, np.zeros((9, 9)))
    assert np.array_equal(agp.mean(fx), f.mean(x)) and np.array_equal(agp.cov(fx), f.cov(x))
    assert np.array_equal(agp.cov(fx, fx2), f.cov(x, x2))
    m, s = agp.marginals(fx)
    assert np.array_equal(m, f.mean(x)) and np.array_equal(s, np.sqrt(f.var(x)))
    X = kf.ColVecs(r.standard_normal((2, 10)))
    fX = agp.FiniteGP(st.atomic(agp.GP(1, kf.SEKernel()), st.GPC()), X, 1e-12)
    S = 100_000
    fh = agp.rand(fX, r.standard_normal((10, S)))
    assert np.abs(fh.mean(1) - agp.mean(fX)).max() < 1e-2
    Sig = (fh - agp.mean(fX)[:, None]) @ (fh - agp.mean(fX)[:, None]).T / S
    assert np.mean(np.abs(Sig - agp.cov(fX))) < 1e-2


def test_sparse_finite_gp():
    """test/gp/sparse_finite_gp.jl:1-42 + README.md:71-78."""
    x, xu = np.arange(0.0, 10.0001, 0.1), np.arange(0.0, 10.5, 1.0)
    f = st.atomic(agp.GP(kf.Matern32Kernel()), st.GPC())
    fx = f(x, 1.0)
    fxu = st.SparseFiniteGP(f(x, 1.0), f(xu, 1e-3))
    assert f(x).noise == 1e-18                                       # default jitter (:13)
    r = np.random.default_rng(12345)
    y = agp.rand(fx, r.standard_normal(len(x)))
    assert st.sparse_logpdf(fxu, y) == agp.elbo(agp.VFE(fxu.finducing), fxu.fobs, y)
    for _ in range(10):
        yy = agp.rand(fx, r.standard_normal(len(x)))
        assert agp.logpdf(fx, yy) > st.sparse_logpdf(fxu, yy)        # ELBO is a strict lower bound
    p = st.sparse_posterior(fxu, y)
    assert p.mean(x).shape == x.shape and np.all(p.var(x) > 0)
    g = st.atomic(agp.GP(kf.SEKernel()), st.GPC())
    xs = np.sort(np.random.default_rng(8).uniform(-5, 5, 60))
    ys = np.random.default_rng(9).standard_normal(60)
    lp, el = agp.logpdf(g(xs, 0.1), ys), agp.elbo(agp.VFE(g(xs, 1e-9)), g(xs, 0.1), ys)
    assert abs(lp - el) < 1e-5 * abs(lp)


def test_faithful_vs_direct_distances():
    """App. A.1: the GEMM-trick distances (reference-faithful) and the direct form (what the HIP
    kernel computes) agree to O(eps |x|^2) -- far inside the 1e-8 budget for smooth kernels."""
    r = rng()
    X, Y = r.standard_normal((8, 300)), r.standard_normal((8, 200))
    for k in (kf.SEKernel(), kf.Matern52Kernel(), kf.Matern32Kernel()):
        assert np.abs(k.matrix(X, Y, True) - k.matrix(X, Y, False)).max() < 1e-13
        assert np.abs(k.matrix(X, None, True) - k.matrix(X, None, False)).max() < 1e-13
    assert np.array_equal(np.diag(kf.SEKernel().matrix(X)), np.ones(300))


def test_oracle_logpdf_gradient_against_finite_differences():
    """Pins oracle.abstractgps.logpdf_gradient_wrt_cov (the checker of the HIP gradient path) with
    central differences of the oracle's own logpdf -- the way the reference validates AD
    (test/test_util.jl:78-96, FiniteDifferences central_fdm)."""
    r = rng()
    x, y = r.standard_normal(40), r.standard_normal(40)

    def lp(s2, c, l, dy=None, dm=0.0):
        f = c * st.stretch(st.atomic(agp.GP(dm, kf.Matern52Kernel()), st.GPC()), 1.0 / l)
        return agp.logpdf(f(x, s2), y if dy is None else y + dy)

    s2, c, l = 0.3, 1.7, 0.8
    f = c * st.stretch(st.atomic(agp.GP(kf.Matern52Kernel()), st.GPC()), 1.0 / l)
    val, alpha, G = agp.logpdf_gradient_wrt_cov(f(x, s2), y)
    assert val == pytest.approx(lp(s2, c, l), rel=1e-13)
    h = 1e-6
    assert np.trace(G) == pytest.approx((lp(s2 + h, c, l) - lp(s2 - h, c, l)) / (2 * h), rel=1e-6)
    assert (G * f.cov(x)).sum() * 2 / c == pytest.approx((lp(s2, c + h, l) - lp(s2, c - h, l)) / (2 * h), rel=1e-6)
    e3 = np.zeros(40)
    e3[3] = h
    assert -alpha[3] == pytest.approx((lp(s2, c, l, e3) - lp(s2, c, l, -e3)) / (2 * h), rel=1e-6)


def test_oracle_logpdf_on_ill_conditioned_covariances_tracks_60_digit_values():
    """tests/golden/illcond_truth.json (mpmath, 60 digits; generator committed next to it): the
    oracle's LAPACK path loses about cond(C) * eps and no more."""
    import json
    import os
    with open(os.path.join(os.path.dirname(__file__), "golden", "illcond_truth.json")) as f:
        cases = json.load(f)["cases"]
    for c in cases:
        x, y, s2 = np.array(c["x"]), np.array(c["y"]), c["noise"]
        lo = agp.logpdf(st.atomic(agp.GP(kf.SEKernel()), st.GPC())(x, s2), y)
        truth = float(c["logpdf"])
        assert abs(lo - truth) / abs(truth) < 50 * 2.2e-16 / s2, (c["N"], s2)


def test_oracle_elbo_gradient_matches_finite_differences():
    """oracle.abstractgps.elbo_gradient_wrt_cov against central differences of an explicit
    matrix restatement of the same bound (which is itself checked against oracle elbo)."""
    import scipy.linalg as sla
    rng = np.random.default_rng(5)
    N, M = 23, 7
    x, z = np.sort(rng.uniform(-2, 2, N)), np.linspace(-1.8, 1.8, M)
    y = rng.standard_normal(N)
    f = st.atomic(agp.GP(0.3, 1.7 * kf.with_lengthscale(kf.Matern52Kernel(), 0.8)), st.GPC())
    sy = 0.05 + rng.random(N)

    def bound(Kzz, Kxz, var, s, yy):
        Lz = np.linalg.cholesky(Kzz)
        A = sla.solve_triangular(Lz, Kxz.T, lower=True) / np.sqrt(s)[None, :]
        Le = np.linalg.cholesky(A @ A.T + np.eye(M))
        d = (yy - 0.3) / np.sqrt(s)
        b = sla.solve_triangular(Le, A @ d, lower=True)
        return (-0.5 * (N * np.log(2 * np.pi) + np.log(s).sum() + 2 * np.log(np.diag(Le)).sum() + d @ d - b @ b)
                - 0.5 * ((var / s).sum() - (A * A).sum()))

    for noise in (sy, 0.1):
        fx, fz = f(x, noise), f(z, 1e-6)
        g = agp.elbo_gradient_wrt_cov(agp.VFE(fz), fx, y)
        Kzz = f.cov(z) + 1e-6 * np.eye(M)
        Kxz = f.cov_cross(f, x, z) if hasattr(f, "cov_cross") else f.cov(x, z)
        var = f.var(x)
        s = agp.noise_diag(noise, N)
        assert abs(bound(Kzz, Kxz, var, s, y) - g["elbo"]) < 1e-9 * abs(g["elbo"])
        h = 1e-6
        for _ in range(6):
            i, j = rng.integers(M), rng.integers(M)
            E = np.zeros((M, M)); E[i, j] += 0.5; E[j, i] += 0.5
            fd = (bound(Kzz + h * E, Kxz, var, s, y) - bound(Kzz - h * E, Kxz, var, s, y)) / (2 * h)
            an = 0.5 * (g["Kzz"][i, j] + g["Kzz"][j, i])
            assert abs(fd - an) < 1e-5 * max(1.0, abs(an)), (fd, an)
            i, j = rng.integers(N), rng.integers(M)
            E = np.zeros((N, M)); E[i, j] = 1.0
            fd = (bound(Kzz, Kxz + h * E, var, s, y) - bound(Kzz, Kxz - h * E, var, s, y)) / (2 * h)
            assert abs(fd - g["Kxz"][i, j]) < 1e-5 * max(1.0, abs(fd)), (fd, g["Kxz"][i, j])
            i = rng.integers(N)
            e = np.zeros(N); e[i] = 1.0
            fd = (bound(Kzz, Kxz, var, s, y + h * e) - bound(Kzz, Kxz, var, s, y - h * e)) / (2 * h)
            assert abs(fd - g["y"][i]) < 1e-5 * max(1.0, abs(fd))
            fd = (bound(Kzz, Kxz, var + h * e, s, y) - bound(Kzz, Kxz, var - h * e, s, y)) / (2 * h)
            assert abs(fd - g["var"][i]) < 1e-5 * max(1.0, abs(fd))
            if np.ndim(noise) == 1:
                fd = (bound(Kzz, Kxz, var, s + h * e, y) - bound(Kzz, Kxz, var, s - h * e, y)) / (2 * h)
                assert abs(fd - g["noise"][i]) < 1e-5 * max(1.0, abs(fd)), (fd, g["noise"][i])
        if np.ndim(noise) == 0:
            fd = (bound(Kzz, Kxz, var, s + h, y) - bound(Kzz, Kxz, var, s - h, y)) / (2 * h)
            assert abs(fd - g["noise"]) < 1e-5 * max(1.0, abs(fd)), (fd, g["noise"])
