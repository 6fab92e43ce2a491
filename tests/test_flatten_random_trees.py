"""Host logic (no GPU): RANDOM Stheno programmes.  The fixed recipes of tests/models.py are the models the reference's
own tests and examples use; this suite composes atoms with +, -, real and function scaling, stretch / shift / periodic /
arbitrary input maps at random (shared sub-trees included -- the case where the reference's recursion re-evaluates a
node exponentially often and the flattener merges paths), builds every programme twice -- with the oracle's literal
restatement of the recursion (oracle/stheno.py: derived_gp.jl:31-60, addition.jl:26-54, product.jl:25-70,
compose.jl:16-28) and with the product's flattener (stheno.jl_amd/flatten.py) -- and compares the dense covariance of
a BlockData over ALL its processes, cross-covariances between two input collections, and the means."""
import numpy as np
import pytest

import models
import np_terms
import oracle.kernelfunctions as okf
import oracle.stheno as ost
import stheno_jl_amd as P


@pytest.fixture(autouse=True)
def _direct_distances(monkeypatch):
    """The oracle's kernel matrices follow Distances.jl (|a|^2 + |b|^2 - 2 a'b): between two VIEWS of one atom at the
    same point that leaves d^2 ~ 1e-16 instead of 0, which Matern-1/2 (exp(-d)) turns into 1e-8.  The flattener is
    compared here on the algebra, so the recursion runs on direct differences like the device
    (tests/test_oracle_reference_properties.py::test_faithful_vs_direct_distances bounds the difference itself)."""
    orig = okf.pairwise_sqeuclidean
    monkeypatch.setattr(okf, "pairwise_sqeuclidean", lambda X, Y=None, faithful=True: orig(X, Y, False))


def _sumsin(x):
    return float(np.sum(np.sin(x)))


def _one_plus_sq(x):
    return float(1.0 + 0.3 * np.sum(np.square(x)))


def _build(api, seed, n_atoms, n_ops, D=1):
    """The same structural random stream for both APIs -> the same programme.  D = 1: scalar inputs, a third of the
    atoms are only ever seen through periodic(...) (every view of one atom must have one input dimension, so the raw
    atom stays out of the programme); D > 1: ColVecs inputs with scalar / diagonal / matrix stretches and vector shifts."""
    rng = np.random.default_rng(seed)
    gpc = api.GPC()
    kernels = [api.SEKernel, api.Matern12Kernel, api.Matern32Kernel, api.Matern52Kernel]
    means = [None, 0.7, _sumsin, _one_plus_sq]
    nodes = []
    for _ in range(n_atoms):
        k = kernels[rng.integers(len(kernels))]()
        if rng.random() < 0.4:
            k = api.with_lengthscale(k, float(0.5 + rng.random()))
        if rng.random() < 0.3:
            k = api.ScaledKernel(k, float(0.5 + rng.random()))
        m = means[rng.integers(len(means))]
        f = api.atomic(api.GP(k) if m is None else api.GP(m, k), gpc)
        if D == 1 and rng.random() < 0.33:
            if rng.random() < 0.5:
                f = api.stretch(f, float(0.5 + rng.random()))
            f = api.periodic(f, float(0.5 + rng.random()))
        nodes.append(f)
    for _ in range(n_ops):
        op = rng.integers(8)
        a = nodes[rng.integers(len(nodes))]
        b = nodes[rng.integers(len(nodes))]
        c = float(np.round(rng.standard_normal() * 2.0, 3)) or 0.5
        if op == 0:
            f = a + b
        elif op == 1:
            f = a - b
        elif op == 2:
            f = c * a
        elif op == 3:
            f = a * c
        elif op == 4:
            f = (_sumsin if rng.random() < 0.5 else _one_plus_sq) * a
        elif op == 5:
            kind = rng.integers(3) if D > 1 else 0
            if kind == 0:
                f = api.stretch(a, float(0.3 + rng.random()))
            elif kind == 1:
                f = api.stretch(a, 0.3 + rng.random(D))
            else:
                f = api.stretch(a, np.eye(D) * 0.8 + 0.3 * rng.standard_normal((D, D)))
        elif op == 6:
            f = api.shift(a, float(rng.standard_normal()) if D == 1 else rng.standard_normal(D))
        else:
            f = api.compose(a, np.cos)
        nodes.append(f)
    return {f"f{i}": f for i, f in enumerate(nodes)}, gpc


@pytest.mark.parametrize("seed", range(40))
def test_random_programme_flattens_to_the_recursions_matrix(seed):
    n_atoms, n_ops = 2 + seed % 3, 4 + seed % 7
    fo, go = _build(models.oracle_api(), seed, n_atoms, n_ops)
    fp, gp = _build(models.product_api(), seed, n_atoms, n_ops)
    names = list(fo)
    rng = np.random.default_rng(10_000 + seed)
    xs = [rng.standard_normal(2 + (i + seed) % 3) for i in range(len(names))]
    xo = ost.BlockData([ost.GPPPInput(k, x) for k, x in zip(names, xs)])
    xp = P.BlockData([P.GPPPInput(k, x) for k, x in zip(names, xs)])
    Fo, Fp = ost.GPPP(fo, go), P.GPPP(fp, gp)
    Ko = Fo.cov(xo)
    spec, _, _ = P.build_spec(Fp, xp)
    Kp = np_terms.dense_from_spec(spec)
    scale = max(1.0, float(np.abs(Ko).max()))
    np.testing.assert_allclose(Kp, Ko, rtol=1e-11, atol=1e-12 * scale)
    np.testing.assert_allclose(P.mean_vector(Fp, xp), Fo.mean(xo), rtol=1e-12, atol=1e-12)
    # exact zeros of the recursion (independent atoms) must be exact zeros of the flattened spec
    assert np.all((Ko == 0.0) <= (Kp == 0.0))
    # the covariance is symmetric and the diagonal path agrees with it
    np.testing.assert_allclose(Kp, Kp.T, rtol=0, atol=1e-12 * scale)
    # cross-covariance between two different collections over a subset of the processes, in another order
    sub = names[::-1][: max(2, len(names) // 2)]
    ys = [rng.standard_normal(1 + (i % 3)) for i in range(len(sub))]
    yo = ost.BlockData([ost.GPPPInput(k, y) for k, y in zip(sub, ys)])
    yp = P.BlockData([P.GPPPInput(k, y) for k, y in zip(sub, ys)])
    specx, _, _ = P.build_spec(Fp, xp, Fp, yp)
    np.testing.assert_allclose(np_terms.dense_from_spec(specx), Fo.cov(xo, yo), rtol=1e-11, atol=1e-12 * scale)


@pytest.mark.parametrize("seed", range(100, 125))
def test_random_programme_on_colvecs(seed):
    D = 2 + seed % 2
    n_atoms, n_ops = 2 + seed % 2, 5 + seed % 5
    fo, go = _build(models.oracle_api(), seed, n_atoms, n_ops, D)
    fp, gp = _build(models.product_api(), seed, n_atoms, n_ops, D)
    names = list(fo)
    rng = np.random.default_rng(20_000 + seed)
    xs = [np.asfortranarray(rng.standard_normal((D, 1 + (i + seed) % 4))) for i in range(len(names))]
    xo = ost.BlockData([ost.GPPPInput(k, okf.ColVecs(x)) for k, x in zip(names, xs)])
    xp = P.BlockData([P.GPPPInput(k, P.ColVecs(x)) for k, x in zip(names, xs)])
    Fo, Fp = ost.GPPP(fo, go), P.GPPP(fp, gp)
    Ko = Fo.cov(xo)
    Kp = np_terms.dense_from_spec(P.build_spec(Fp, xp)[0])
    scale = max(1.0, float(np.abs(Ko).max()))
    np.testing.assert_allclose(Kp, Ko, rtol=1e-11, atol=1e-12 * scale)
    np.testing.assert_allclose(P.mean_vector(Fp, xp), Fo.mean(xo), rtol=1e-12, atol=1e-12)
    assert np.all((Ko == 0.0) <= (Kp == 0.0))


# ---- random kernel expressions (KernelFunctions composites: ScaledKernel, KernelSum, TransformedKernel) ------------
def _random_kernel(api, rng, depth, periodic_ok):
    """A random kernel expression; at most one PeriodicTransform on the way from the raw 1-D point to a leaf."""
    leaves = [api.SEKernel, api.Matern12Kernel, api.Matern32Kernel, api.Matern52Kernel, api.WhiteKernel]
    op = rng.integers(7) if depth > 0 else 6
    if op == 0:
        return api.ScaledKernel(_random_kernel(api, rng, depth - 1, periodic_ok), float(0.2 + 2 * rng.random()))
    if op == 1:
        n = 2 + rng.integers(2)
        return api.KernelSum([_random_kernel(api, rng, depth - 1, periodic_ok) for _ in range(n)])
    if op == 2:
        return api.with_lengthscale(_random_kernel(api, rng, depth - 1, periodic_ok), float(0.4 + 2 * rng.random()))
    if op == 3:
        return api.TransformedKernel(_random_kernel(api, rng, depth - 1, periodic_ok), api.ScaleTransform(float(0.3 + rng.random())))
    if op == 4 and periodic_ok:
        # everything below reads the 2-D image of the point: no second PeriodicTransform there
        return api.TransformedKernel(_random_kernel(api, rng, depth - 1, False), api.PeriodicTransform(float(0.3 + rng.random())))
    if op == 5:
        return api.ConstantKernel(float(0.1 + rng.random()))
    return leaves[rng.integers(len(leaves))]()


@pytest.mark.parametrize("seed", range(200, 260))
def test_random_kernel_expression_expands_to_the_same_matrix(seed):
    ko = _random_kernel(models.oracle_api(), np.random.default_rng(seed), 1 + seed % 4, True)
    kp = _random_kernel(models.product_api(), np.random.default_rng(seed), 1 + seed % 4, True)
    rng = np.random.default_rng(30_000 + seed)
    x = rng.standard_normal(6)
    x[4] = x[1]                                  # a repeated point: WhiteKernel's delta off the diagonal
    z = np.concatenate([rng.standard_normal(3), x[:2]])
    f = P.atomic(P.GP(kp), P.GPC())
    Ko = okf.kernelmatrix(ko, x)
    Kp = np_terms.dense_from_spec(P.build_spec(f, x)[0])
    scale = max(1.0, float(np.abs(Ko).max()))
    np.testing.assert_allclose(Kp, Ko, rtol=1e-11, atol=1e-12 * scale)
    Kxo = okf.kernelmatrix(ko, x, z)
    Kxp = np_terms.dense_from_spec(P.build_spec(f, x, f, z)[0])
    np.testing.assert_allclose(Kxp, Kxo, rtol=1e-11, atol=1e-12 * scale)


# ---- the whole host mirror on random programmes (library = the NumPy double of its C-ABI, tests/np_capi.py) ---------
@pytest.mark.parametrize("seed", range(300, 325))
def test_random_programme_logpdf_posterior_and_gradient_through_the_host_mirror(seed, monkeypatch):
    import np_capi
    import oracle.abstractgps as oagp
    np_capi.install(monkeypatch)
    n_atoms, n_ops = 2 + seed % 3, 3 + seed % 6
    fo, go = _build(models.oracle_api(), seed, n_atoms, n_ops)
    fp, gp = _build(models.product_api(), seed, n_atoms, n_ops)
    names = list(fo)
    rng = np.random.default_rng(40_000 + seed)
    xs = [rng.standard_normal(3 + (i + seed) % 4) for i in range(len(names))]
    xo = ost.BlockData([ost.GPPPInput(k, x) for k, x in zip(names, xs)])
    xp = P.BlockData([P.GPPPInput(k, x) for k, x in zip(names, xs)])
    Fo, Fp = ost.GPPP(fo, go), P.GPPP(fp, gp)
    N = sum(len(x) for x in xs)
    y = rng.standard_normal(N)
    noise = 0.3 + rng.random(N)
    lo, lp = oagp.logpdf(Fo(xo, noise), y), P.logpdf(Fp(xp, noise), y)
    assert abs(lp - lo) <= 1e-9 * max(1.0, abs(lo))
    k = names[-1]
    t = rng.standard_normal(4)
    po, pp = oagp.posterior(Fo(xo, noise), y), P.posterior(Fp(xp, noise), y)
    mo, vo = po.mean_and_var(ost.GPPPInput(k, t))
    mp, vp = pp.mean_and_var(P.GPPPInput(k, t))
    np.testing.assert_allclose(mp, mo, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(vp, vo, rtol=1e-8, atol=1e-9)
    # the gradient records: d logpdf / d (a common factor on every coefficient) = sum_t coef_t d_coef_t, against a
    # finite difference of the product's own logpdf with the whole covariance scaled
    g = P.logpdf_and_gradient(Fp(xp, noise), y)
    lhs = sum(r["coef"] * r["d_coef"] for r in g["terms"])
    _, alpha, Gm = oagp.logpdf_gradient_wrt_cov(Fo(xo, noise), y)
    rhs = float((Gm * Fo.cov(xo)).sum())          # d logpdf / d s at s = 1 for C = s K + Sigma_y
    assert abs(lhs - rhs) <= 1e-7 * max(1.0, abs(rhs)), (lhs, rhs)
    np.testing.assert_allclose(g["y"], -alpha, rtol=1e-8, atol=1e-9)
