"""Host logic (no GPU): the product's flattener (tree -> kernel terms, stheno.jl_amd/flatten.py)
must describe exactly the matrix the reference's recursion (restated literally in
oracle/stheno.py from src/gp/derived_gp.jl:31-60 etc.) evaluates.  The terms are evaluated by
the NumPy test double tests/np_terms.py, so this isolates flattening from the HIP kernels."""
import itertools

import numpy as np
import pytest

import models
import np_terms
import oracle.abstractgps as oagp
import oracle.kernelfunctions as okf
import oracle.stheno as ost
import stheno_jl_amd as P

RTOL = 1e-12


def _inputs_1d(rng, n):
    return rng.standard_normal(n)


def _pair(recipe):
    fo, go = recipe(models.oracle_api())
    fp, gp = recipe(models.product_api())
    return fo, go, fp, gp


def _mean_close(a, b):
    np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("recipe", models.RECIPES_1D + models.RECIPES_1D_CPU_ONLY, ids=lambda r: r.__name__)
def test_gppp_blockdata_1d(recipe):
    rng = np.random.default_rng(123456)
    fo, go, fp, gp = _pair(recipe)
    names = list(fo)
    sizes = [3 + (i % 4) for i in range(len(names))]
    xs = [_inputs_1d(rng, n) for n in sizes]
    xo = ost.BlockData([ost.GPPPInput(k, x) for k, x in zip(names, xs)])
    xp = P.BlockData([P.GPPPInput(k, x) for k, x in zip(names, xs)])
    Fo, Fp = ost.GPPP(fo, go), P.GPPP(fp, gp)
    Ko = Fo.cov(xo)
    spec, _, _ = P.build_spec(Fp, xp)
    Kp = np_terms.dense_from_spec(spec)
    np.testing.assert_allclose(Kp, Ko, rtol=RTOL, atol=1e-13)
    _mean_close(P.mean_vector(Fp, xp), Fo.mean(xo))
    # exact zeros between independent processes must be *no terms*, not small numbers
    assert np.array_equal(Kp == 0.0, np.abs(Ko) == 0.0) or np.all((Ko == 0) <= (Kp == 0))
    # cross-covariance against a different BlockData (subset, different sizes)
    sub = names[::2]
    ys = [_inputs_1d(rng, 2 + i) for i in range(len(sub))]
    yo = ost.BlockData([ost.GPPPInput(k, y) for k, y in zip(sub, ys)])
    yp = P.BlockData([P.GPPPInput(k, y) for k, y in zip(sub, ys)])
    Kxo = Fo.cov(xo, yo)
    specx, _, _ = P.build_spec(Fp, xp, Fp, yp)
    np.testing.assert_allclose(np_terms.dense_from_spec(specx), Kxo, rtol=RTOL, atol=1e-13)


@pytest.mark.parametrize("recipe", models.RECIPES_ND, ids=lambda r: r.__name__)
def test_pairs_colvecs(recipe):
    rng = np.random.default_rng(20260925)
    D = 3
    fo, go, fp, gp = _pair(recipe)
    X1 = np.asfortranarray(rng.standard_normal((D, 5)))
    X2 = np.asfortranarray(rng.standard_normal((D, 4)))
    tested = 0
    for a, b in itertools.product(list(fo), repeat=2):
        try:
            Ko = ost.cov4(fo[a], fo[b], okf.ColVecs(X1), okf.ColVecs(X2))
        except Exception:
            # dimension-incompatible views of one atom (e.g. select(f, 1) against f): the product
            # must refuse as well
            with pytest.raises(Exception):
                s, _, _ = P.build_spec(fp[a], P.ColVecs(X1), fp[b], P.ColVecs(X2))
                np_terms.dense_from_spec(s)
            continue
        s, _, _ = P.build_spec(fp[a], P.ColVecs(X1), fp[b], P.ColVecs(X2))
        np.testing.assert_allclose(np_terms.dense_from_spec(s), Ko, rtol=RTOL, atol=1e-13, err_msg=f"{a},{b}")
        _mean_close(P.mean_vector(fp[a], P.ColVecs(X1)), fo[a].mean(okf.ColVecs(X1)))
        tested += 1
    assert tested >= len(fo)


def test_term_merging_counts():
    """f4 = f1 + (f1 + f2): the diagonal block is 4 k1 + k2.  Here k1 and k2 are both SE on the
    same inputs, so everything merges into ONE term with coefficient 5; with different leaf
    kernels the two atoms stay separate (coefficients 4 and 1)."""
    fp, gp = models.correlated_sums(models.product_api())
    x = np.linspace(-1, 1, 6)
    s, _, _ = P.build_spec(fp["f4"], x)
    terms = np_terms.spec_terms(s)
    assert len(terms) == 1 and terms[0][5] == 5.0
    g = P.GPC()
    a, b = P.atomic(P.GP(P.SEKernel()), g), P.atomic(P.GP(P.Matern32Kernel()), g)
    s, _, _ = P.build_spec(a + (a + b), x)
    assert sorted(t[5] for t in np_terms.spec_terms(s)) == [1.0, 4.0]


def test_independent_blocks_have_no_terms():
    fp, gp = models.gppp_docstring(models.product_api())
    F = P.GPPP(fp, gp)
    x = P.BlockData([P.GPPPInput("f1", np.arange(3.0)), P.GPPPInput("f2", np.arange(4.0))])
    s, _, _ = P.build_spec(F, x)
    pairs = {(t[0], t[1]) for t in np_terms.spec_terms(s)}
    assert pairs == {(0, 0), (1, 1)}


def test_generic_tuple_vector_regroups_like_reference():
    """gppp.jl:32-43: a plain vector of (key, value) is regrouped by unique key."""
    fo, go, fp, gp = _pair(models.toy_gppp)
    Fo, Fp = ost.GPPP(fo, go), P.GPPP(fp, gp)
    items = [("f2", 0.3), ("f3", -1.0), ("f2", 0.7), ("f1", 0.1), ("f3", 2.0)]
    Ko = Fo.cov(items)
    s, _, _ = P.build_spec(Fp, items)
    np.testing.assert_allclose(np_terms.dense_from_spec(s), Ko, rtol=RTOL, atol=1e-14)
    _mean_close(P.mean_vector(Fp, items), Fo.mean(items))


def test_nested_gppp():
    """test/gaussian_process_probabilistic_programme.jl:107-120."""
    fo, go, fp, gp = _pair(models.toy_gppp)
    Fo, Fp = ost.GPPP(fo, go), P.GPPP(fp, gp)
    g2o, g2p = ost.GPC(), P.GPC()
    f1o, f1p = ost.atomic(Fo, g2o), P.atomic(Fp, g2p)
    Oo = ost.GPPP({"f1": f1o, "f2": 5 * f1o}, g2o)
    Op = P.GPPP({"f1": f1p, "f2": 5 * f1p}, g2p)
    x0, x1 = np.array([0.1, 0.5, -0.3]), np.array([1.0, -1.0])
    xo = ost.BlockData([ost.GPPPInput("f1", ost.GPPPInput("f1", x0)), ost.GPPPInput("f2", ost.GPPPInput("f3", x1))])
    xp = P.BlockData([P.GPPPInput("f1", P.GPPPInput("f1", x0)), P.GPPPInput("f2", P.GPPPInput("f3", x1))])
    Ko = Oo.cov(xo)
    s, _, _ = P.build_spec(Op, xp)
    np.testing.assert_allclose(np_terms.dense_from_spec(s), Ko, rtol=RTOL, atol=1e-14)
    _mean_close(P.mean_vector(Op, xp), Oo.mean(xo))


def test_nested_gppp_with_inner_inputs_spanning_several_inner_processes():
    """The outer process 5 * atomic(inner programme) indexed (a) with a BlockData of inner GPPPInputs and (b) with a generic
    vector of (key, value) pairs, which the inner programme regroups by key (gppp.jl:32-43: the element order changes --
    the same way in the oracle's recursion and in the flattening, which regroups before it walks any path)."""
    fo, go, fp, gp = _pair(models.toy_gppp)
    Fo, Fp = ost.GPPP(fo, go), P.GPPP(fp, gp)
    g2o, g2p = ost.GPC(), P.GPC()
    f1o, f1p = ost.atomic(Fo, g2o), P.atomic(Fp, g2p)
    Oo = ost.GPPP({"f1": f1o, "f2": 5 * f1o}, g2o)
    Op = P.GPPP({"f1": f1p, "f2": 5 * f1p}, g2p)
    x0, x1 = np.array([0.1, 0.5, -0.3]), np.array([1.0, -1.0])
    pairs = [("f1", 0.1), ("f3", 1.0), ("f1", 0.5), ("f2", -0.7)]
    for inner in (lambda api: api.BlockData([api.GPPPInput("f1", x0), api.GPPPInput("f3", x1)]), lambda api: pairs):
        xo = ost.BlockData([ost.GPPPInput("f1", ost.GPPPInput("f1", x0)), ost.GPPPInput("f2", inner(ost))])
        xp = P.BlockData([P.GPPPInput("f1", P.GPPPInput("f1", x0)), P.GPPPInput("f2", inner(P))])
        Ko = Oo.cov(xo)
        s, _, _ = P.build_spec(Op, xp)
        assert Ko.shape[0] == len(xp)
        np.testing.assert_allclose(np_terms.dense_from_spec(s), Ko, rtol=RTOL, atol=1e-14)
        _mean_close(P.mean_vector(Op, xp), Oo.mean(xo))


def test_gpc_mismatch_is_rejected():
    a, _ = models.gppp_docstring(models.product_api())
    b, _ = models.gppp_docstring(models.product_api())
    with pytest.raises(AssertionError):
        a["f1"] + b["f1"]
    with pytest.raises(AssertionError):
        P.build_spec(a["f1"], np.arange(3.0), b["f1"], np.arange(3.0))
    with pytest.raises(ValueError):
        a["f1"] * a["f2"]


def test_chain_rule_of_input_gradients_through_warps_and_kernel_scales():
    """flatten.chain_input_gradients (the host half of the input-gradient path): for a model whose
    terms read stretched / selected / periodic / shifted / length-scaled views of the block inputs,
    the chained gradient of Phi(x) = sum_k <W_k, spec.inputs[k](x)> must match central differences
    (Phi is linear in the spec inputs, so its gradient w.r.t. them is W_k itself)."""
    from stheno_jl_amd.flatten import chain_input_gradients
    rng = np.random.default_rng(31)
    gpc = P.GPC()
    a = P.atomic(P.GP(P.with_lengthscale(P.SEKernel(), 0.7)), gpc)
    b = P.atomic(P.GP(P.Matern52Kernel()), gpc)
    f1 = P.stretch(a, 1.9)                                        # 3-D inputs, scalar stretch + kernel scale
    f2 = P.stretch(P.shift(a, np.array([0.3, -0.2, 0.1])), np.array([0.5, 2.0, 1.5]))   # diag stretch o shift
    f3 = P.periodic(b, 0.8)                                       # 1-D -> [cos, sin]
    f4 = P.select(P.stretch(a, 0.9), [2, 0, 1])                   # row permutation of a 3-D input
    F = P.GPPP({"f1": f1, "f2": f2, "f3": f3, "f4": f4, "f5": f1 + f4}, gpc)
    xs = [P.ColVecs(rng.standard_normal((3, 5))), P.ColVecs(rng.standard_normal((3, 4))), rng.standard_normal(6),
          P.ColVecs(rng.standard_normal((3, 3))), P.ColVecs(rng.standard_normal((3, 4)))]
    names = ["f1", "f2", "f3", "f4", "f5"]

    def spec_of(mats):
        blocks = [P.GPPPInput(k, (P.ColVecs(m) if m.shape[0] == 3 else m.reshape(-1))) for k, m in zip(names, mats)]
        return P.build_spec(F, P.BlockData(blocks))[0]

    mats = [np.array(P.inputs.as_matrix(x), dtype=float) for x in xs]
    spec = spec_of(mats)
    W = [rng.standard_normal(np.asarray(v).shape) for v in spec.inputs]

    def phi(ms):
        sp = spec_of(ms)
        assert len(sp.inputs) == len(W)
        return sum(float((w * np.asarray(v)).sum()) for w, v in zip(W, sp.inputs))

    rows, cols = chain_input_gradients(spec, W)
    assert cols is rows and len(rows) == len(xs)
    h = 1e-6
    for I, m in enumerate(mats):
        assert rows[I].shape == m.shape
        for (d, i) in itertools.product(range(m.shape[0]), range(m.shape[1])):
            mp_, mn_ = [q.copy() for q in mats], [q.copy() for q in mats]
            mp_[I][d, i] += h
            mn_[I][d, i] -= h
            fd = (phi(mp_) - phi(mn_)) / (2 * h)
            assert abs(rows[I][d, i] - fd) <= 1e-7 * max(1.0, abs(fd)), (I, d, i, rows[I][d, i], fd)


def test_gradient_records_fold_each_mirror_term_into_its_own_lower_term():
    """f3 = f1 + sin * f1 over two blocks gives four terms per block pair that differ only
    in their row / column scale vectors; the mirror of (rs = a, cs = b) in pair (I, J) is (rs = b,
    cs = a) in pair (J, I).  Checked numerically: the mirror term's matrix is the transpose of the
    lower term's, and every upper term is folded exactly once."""
    import np_terms
    from stheno_jl_amd import finite_gp as fg
    rng = np.random.default_rng(11)
    gpc = P.GPC()
    f1 = P.atomic(P.GP(P.SEKernel()), gpc)
    f3 = f1 + (lambda x: float(np.sum(np.sin(x)))) * f1
    F = P.GPPP({"f1": f1, "f3": f3}, gpc)
    x = P.BlockData([P.GPPPInput("f3", rng.standard_normal(7)), P.GPPPInput("f3", rng.standard_normal(5)),
                     P.GPPPInput("f1", rng.standard_normal(4))])
    spec = P.build_spec(F, x)[0]
    terms = np_terms.spec_terms(spec)
    nt = len(terms)
    assert nt == 4 * 4 + 2 * 2 * 2 + 1      # (f3,f3) pairs: 4 terms each; (f3,f1): 2; (f1,f1): 1
    gc = np.arange(1.0, nt + 1.0)
    gs = 100.0 + gc
    recs = fg._term_records(spec, gc, gs, True)

    def mat(t):
        I, J, kind, ri, ci, coef, param, rs, cs = terms[t]
        X, Y = spec.inputs[ri], spec.inputs[ci]
        d2 = (X[0][:, None] - Y[0][None, :]) ** 2
        blk = coef * np_terms._kern(kind, d2, param)
        if rs is not None:
            blk = rs[:, None] * blk
        if cs is not None:
            blk = blk * cs[None, :]
        return blk

    folded = set()
    for r in recs:
        t = r["t"]
        if r["I"] == r["J"]:
            assert r["mirror_t"] is None and r["d_coef"] == gc[t]
            continue
        m = r["mirror_t"]
        assert m is not None and m not in folded
        folded.add(m)
        assert np.allclose(mat(m), mat(t).T, rtol=1e-14, atol=0)   # (a k) b vs (b k) a: rounding only
        assert r["d_coef"] == gc[t] + gc[m] and r["d_inscale"] == gs[t] + gs[m]
    n_upper = sum(1 for (I, J, *_) in terms if I < J)
    assert len(folded) == n_upper


def test_plain_gp_indexed_with_blockdata_is_an_ordinary_vector():
    """BlockData is an AbstractVector for any GP (input_collection_types.jl:61-95), not only
    for cross / GPPP nodes: cov(f, BlockData([a, b])) == cov(f, vcat(a, b))."""
    import np_terms
    rng = np.random.default_rng(5)
    f = P.stretch(P.atomic(P.GP(np.sin, P.Matern32Kernel()), P.GPC()), 0.7)
    a, b = rng.standard_normal(6), rng.standard_normal(9)
    Kb = np_terms.dense_from_spec(P.build_spec(f, P.BlockData([a, b]))[0])
    Kv = np_terms.dense_from_spec(P.build_spec(f, np.concatenate([a, b]))[0])
    assert np.allclose(Kb, Kv, rtol=0, atol=1e-15)
    assert np.array_equal(P.mean_vector(f, P.BlockData([a, b])), P.mean_vector(f, np.concatenate([a, b])))
    Kc = np_terms.dense_from_spec(P.build_spec(f, P.BlockData([a, b]), f, a)[0])
    assert np.allclose(Kc, Kv[:, :6], rtol=0, atol=1e-15)


def test_sparse_finite_gp_dispatch_matches_reference():
    """sparse_finite_gp.jl:37-43: mean(f) = mean(f.fobs); cov(f) raises the explicit dense-covariance
    error.  (mean needs no GPU: prior means are host-side.)"""
    import pytest
    rng = np.random.default_rng(6)
    f = P.atomic(P.GP(np.cos, P.SEKernel()), P.GPC())
    x, z = rng.standard_normal(11), rng.standard_normal(4)
    sf = P.SparseFiniteGP(f(x, 0.1), f(z, 1e-6))
    assert np.array_equal(P.mean(sf), np.cos(x))
    with pytest.raises(RuntimeError, match="covariance matrix of a sparse GP"):
        P.cov(sf)


def test_kernel_level_transform_chain_and_its_vjp():
    """TransformedKernel chains (ScaleTransform / PeriodicTransform / with_lengthscale): composition order
    (outer transforms act on the raw input first), merging of adjacent scalings, and the chain rule used
    by the input gradients, against finite differences."""
    from stheno_jl_amd import kernels as K
    k = P.with_lengthscale(P.SEKernel() @ P.PeriodicTransform(0.7), 2.0) @ P.ScaleTransform(3.0)
    (kind, coef, param, chain), = k.leaf_terms()
    assert chain == (("scale", 1.5), ("periodic", 0.7))           # x -> 3 x -> x / 2 -> periodic
    rng = np.random.default_rng(4)
    X = rng.standard_normal((1, 9))
    Y = K.apply_chain(chain, X)
    t = 2 * np.pi * 0.7 * 1.5 * X
    assert np.allclose(Y, np.vstack([np.sin(t), np.cos(t)]), rtol=0, atol=1e-15)
    G = rng.standard_normal(Y.shape)
    g = K.chain_vjp(chain, X, G)
    h = 1e-6
    fd = np.array([(np.sum(G * K.apply_chain(chain, X + h * np.eye(1, 9, j))) -
                    np.sum(G * K.apply_chain(chain, X - h * np.eye(1, 9, j)))) / (2 * h) for j in range(9)])
    assert np.allclose(g.ravel(), fd, rtol=1e-7, atol=1e-8)
    assert K.chain_scale(P.with_lengthscale(P.SEKernel(), 4.0).leaf_terms()[0][3]) == 0.25
    assert K.chain_scale(chain) is None


def test_float32_inputs_are_tagged_for_the_fp32_path():
    """Float32 type stability (reference: test/gp/util.jl:76-88) starts on the host: input collections given in
    Float32 keep that element type as a tag (the host algebra itself runs on the exact fp64 copy)."""
    from stheno_jl_amd.inputs import eltype
    X32 = np.ones((2, 5), dtype=np.float32)
    assert eltype(P.ColVecs(X32)) == np.float32 and eltype(P.ColVecs(X32.astype(np.float64))) == np.float64
    assert P.ColVecs(X32).X.dtype == np.float64
    assert eltype(np.zeros(3, dtype=np.float32)) == np.float32 and eltype(np.zeros(3)) == np.float64
    assert eltype(P.GPPPInput("f1", np.zeros(3, dtype=np.float32))) == np.float32
    assert eltype(P.GPPPInput("f1", P.ColVecs(X32))) == np.float32
    mixed = P.BlockData([P.GPPPInput("f1", P.ColVecs(X32)), P.GPPPInput("f2", P.ColVecs(X32.astype(np.float64)))])
    assert eltype(mixed) == np.float64
    assert eltype(P.BlockData([P.GPPPInput("f1", P.ColVecs(X32)), P.GPPPInput("f2", P.ColVecs(X32))])) == np.float32


def test_function_scales_remember_their_factors():
    """sigma(x) * f: the path's row-scale vector is the product of its factors and names them (node, inputs,
    values) -- what the chain rule of the scale gradients walks (finite_gp._scale_records)."""
    import stheno_jl_amd as P
    rng = np.random.default_rng(4)
    x1, x2 = rng.standard_normal(11), rng.standard_normal(7)
    gpc = P.GPC()
    f1 = P.atomic(P.GP(P.SEKernel()), gpc)
    s1 = lambda x: 1.0 + 0.5 * float(np.sum(np.sin(x)))
    s2 = lambda x: float(np.exp(0.1 * np.sum(x)))
    g1 = s1 * f1
    h = s2 * g1
    F = P.GPPP({"f1": f1, "g1": g1, "h": h}, gpc)
    spec, _, _ = P.build_spec(F, P.BlockData([P.GPPPInput("h", x1), P.GPPPInput("g1", x2)]))
    vecs = {id(r): r for r in spec.term_row_scale if r is not None}
    assert len(vecs) == 2
    for r in vecs.values():
        prod = np.ones(len(r))
        for node, x, vals in r.factors:
            assert node in (g1, h)
            prod = prod * vals
        assert np.allclose(np.asarray(r), prod)
    lens = sorted((len(r), len(r.factors)) for r in vecs.values())
    assert lens == [(7, 1), (11, 2)]                 # g1 at x2: one factor; h at x1: s2 and the nested s1
