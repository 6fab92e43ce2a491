"""RANDOM Stheno programmes on the device.  tests/test_flatten_random_trees.py composes atoms with +, -, scalings, stretches,
shifts, periodic and arbitrary input maps at random (shared sub-trees included) and compares the flattener with the oracle's
literal recursion (oracle/stheno.py: derived_gp.jl:31-60, addition.jl:26-54, product.jl:25-70, compose.jl:16-28) on the
NumPy double of the C-ABI.  Here the same programmes go through libsthenomi.so at sizes that span several 128-tiles with
ragged block boundaries: covariance and cross-covariance assembly (kernelmatrix.hip), logpdf with diagonal noise, posterior
moments of one process given all, and the gradient records -- against the recursion's dense matrices."""
import numpy as np
import pytest

import models
import oracle.abstractgps as oagp
import oracle.kernelfunctions as okf
import oracle.stheno as ost
import stheno_jl_amd as P
from test_flatten_random_trees import _build

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _direct_distances(monkeypatch):
    # as tests/test_flatten_random_trees.py: the recursion on direct differences, like the device
    orig = okf.pairwise_sqeuclidean
    monkeypatch.setattr(okf, "pairwise_sqeuclidean", lambda X, Y=None, faithful=True: orig(X, Y, False))


def _programme(seed, D):
    n_atoms, n_ops = 2 + seed % 3, 4 + seed % 6
    fo, go = _build(models.oracle_api(), seed, n_atoms, n_ops, D)
    fp, gp = _build(models.product_api(), seed, n_atoms, n_ops, D)
    return list(fo), ost.GPPP(fo, go), P.GPPP(fp, gp)


def _inputs(rng, names, D, lo, hi):
    if D == 1:
        xs = [rng.standard_normal(int(rng.integers(lo, hi))) for _ in names]
        return (ost.BlockData([ost.GPPPInput(k, x) for k, x in zip(names, xs)]),
                P.BlockData([P.GPPPInput(k, x) for k, x in zip(names, xs)]), xs)
    xs = [np.asfortranarray(rng.standard_normal((D, int(rng.integers(lo, hi))))) for _ in names]
    return (ost.BlockData([ost.GPPPInput(k, okf.ColVecs(x)) for k, x in zip(names, xs)]),
            P.BlockData([P.GPPPInput(k, P.ColVecs(x)) for k, x in zip(names, xs)]), xs)


@pytest.mark.parametrize("seed", list(range(500, 512)) + list(range(600, 606)))
def test_random_programme_covariances_on_the_device(seed):
    D = 1 if seed < 600 else 2 + seed % 2
    names, Fo, Fp = _programme(seed, D)
    rng = np.random.default_rng(50_000 + seed)
    xo, xp, _ = _inputs(rng, names, D, 20, 140)     # 5 .. 15 processes: a few hundred to ~1500 points, ragged blocks
    Ko = Fo.cov(xo)
    Kp = P.prior_cov(Fp, xp)
    scale = max(1.0, float(np.abs(Ko).max()))
    np.testing.assert_allclose(Kp, Ko, rtol=1e-10, atol=1e-11 * scale)
    assert np.array_equal(Kp, Kp.T)                          # one triangle built, mirrored: exactly symmetric
    assert np.all((Ko == 0.0) <= (Kp == 0.0))                # independent atoms: exact zeros stay exact zeros
    np.testing.assert_allclose(P.mean_vector(Fp, xp), Fo.mean(xo), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(P.prior_var(Fp, xp), np.diag(Ko), rtol=1e-10, atol=1e-11 * scale)
    sub = names[::-1][: max(2, len(names) // 2)]
    yo, yp, _ = _inputs(rng, sub, D, 3, 50)
    np.testing.assert_allclose(P.prior_cov(Fp, xp, yp), Fo.cov(xo, yo), rtol=1e-10, atol=1e-11 * scale)


@pytest.mark.parametrize("seed", list(range(520, 530)) + list(range(620, 624)))
def test_random_programme_logpdf_posterior_and_gradient_on_the_device(seed):
    D = 1 if seed < 600 else 2 + seed % 2
    names, Fo, Fp = _programme(seed, D)
    rng = np.random.default_rng(60_000 + seed)
    xo, xp, xs = _inputs(rng, names, D, 15, 90)
    N = sum(x.shape[-1] for x in xs)
    y = rng.standard_normal(N)
    noise = 0.3 + rng.random(N)
    lo, lp = oagp.logpdf(Fo(xo, noise), y), P.logpdf(Fp(xp, noise), y)
    assert abs(lp - lo) <= 1e-9 * max(1.0, abs(lo))
    k = names[-1]
    t = rng.standard_normal(40) if D == 1 else np.asfortranarray(rng.standard_normal((D, 40)))
    to = ost.GPPPInput(k, t if D == 1 else okf.ColVecs(t))
    tp = P.GPPPInput(k, t if D == 1 else P.ColVecs(t))
    po, pp = oagp.posterior(Fo(xo, noise), y), P.posterior(Fp(xp, noise), y)
    mo, vo = po.mean_and_var(to)
    mp, vp = pp.mean_and_var(tp)
    np.testing.assert_allclose(mp, mo, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(vp, vo, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(pp.cov(tp), po.cov(to), rtol=1e-8, atol=1e-9)
    # gradient records: d logpdf / d (a common factor on every coefficient) = sum_t coef_t d_coef_t = <G, K>
    g = P.logpdf_and_gradient(Fp(xp, noise), y)
    lhs = sum(r["coef"] * r["d_coef"] for r in g["terms"])
    _, alpha, Gm = oagp.logpdf_gradient_wrt_cov(Fo(xo, noise), y)
    rhs = float((Gm * Fo.cov(xo)).sum())
    assert abs(lhs - rhs) <= 1e-7 * max(1.0, abs(rhs)), (lhs, rhs)
    np.testing.assert_allclose(g["y"], -alpha, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(g["noise"], np.diag(Gm), rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("seed", list(range(540, 552)) + list(range(640, 644)))
def test_random_programme_structural_zeros_change_no_bit(seed, monkeypatch):
    """Random programmes have random independence structure (atoms that never meet, sums that couple some of them): the
    factorisation's structural-zero skipping must not change a bit of logpdf, of the posterior moments or of a
    draw against the dense schedule, on the launch-based and on the dataflow schedule -- whatever the fill-in pattern the
    random block order produces."""
    from test_gpu_fused_potrf import _ctx, _with_ctx
    D = 1 if seed < 600 else 2 + seed % 2
    names, Fo, Fp = _programme(seed, D)
    rng = np.random.default_rng(70_000 + seed)
    xo, xp, xs = _inputs(rng, names, D, 100, 400)         # 5 .. 15 processes of 100 .. 400 points: block boundaries inside tiles
    N = sum(x.shape[-1] for x in xs)
    y = rng.standard_normal(N)
    noise = 0.3 + rng.random(N)
    Z = np.asfortranarray(rng.standard_normal((N, 2)))
    t = rng.standard_normal(30) if D == 1 else np.asfortranarray(rng.standard_normal((D, 30)))
    tp = P.GPPPInput(names[-1], t if D == 1 else P.ColVecs(t))

    def run():
        fx = Fp(xp, noise)
        m, v = P.posterior(fx, y).mean_and_var(tp)
        return dict(lp=np.array([P.logpdf(fx, y)]), m=np.asarray(m), v=np.asarray(v), r=np.asarray(P.rand(None, fx, 2, Z=Z)))

    ref = None
    for env in (dict(SGP_DATAFLOW=0, SGP_STRUCT_ZEROS=0), dict(SGP_DATAFLOW=0, SGP_STRUCT_ZEROS=1),
                dict(SGP_DATAFLOW=1, SGP_DF_FAT_MAX_N=0, SGP_STRUCT_ZEROS=1), dict(SGP_DATAFLOW=1, SGP_DF_FAT_MAX_N=1 << 30, SGP_STRUCT_ZEROS=1)):
        ctx = _ctx(monkeypatch, 11, **env)
        out = _with_ctx(ctx, run)
        ctx.close()
        if ref is None:
            ref = out
        for k in ref:
            assert np.array_equal(ref[k], out[k]), (seed, env, k)
    lo = oagp.logpdf(Fo(xo, noise), y)
    assert abs(ref["lp"][0] - lo) <= 1e-9 * max(1.0, abs(lo))
