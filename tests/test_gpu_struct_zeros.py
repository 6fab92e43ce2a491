"""Structural zeros (round 4; include/sthenomi.h: sgp_ctx_factor_work, csrc/capi.hip: sz_build).

A Stheno programme with independent components has EXACT zero blocks in its covariance -- cross.jl / the flattener emit no
term between two processes that share no atom -- and so has its Cholesky factor.  The reference's LAPACK path multiplies
them out; the factorisation here derives the tile-level pattern of the factor (symbolic factorisation, fill-in included) and
skips every tile product with a structurally zero operand.  The skipped products are exact zeros, so NOTHING may change:
every operator built on the factorisation must come out bit-identical with the skipping on and off, on every schedule, and
equal to the oracle's dense computation; and the work counter must show that something was skipped."""
import numpy as np
import pytest

import oracle.abstractgps as oagp
import oracle.stheno as ost
import stheno_jl_amd as P
from test_gpu_fused_potrf import _ctx, _operators, _with_ctx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("schedule", ["launches", "dataflow", "dataflow-fat"])
@pytest.mark.parametrize("N", [700, 3300, 5200])
def test_every_operator_is_bit_identical_with_and_without_the_skipping(monkeypatch, N, schedule):
    env = {"launches": dict(SGP_DATAFLOW=0), "dataflow": dict(SGP_DATAFLOW=1, SGP_DF_FAT_MAX_N=0),
           "dataflow-fat": dict(SGP_DATAFLOW=1, SGP_DF_FAT_MAX_N=1 << 30)}[schedule]
    off = _ctx(monkeypatch, 11, SGP_STRUCT_ZEROS=0, **env)
    ref, _ = _with_ctx(off, lambda: _operators(N))
    e0, d0 = off.factor_work()
    on = _ctx(monkeypatch, 11, SGP_STRUCT_ZEROS=1, **env)
    for rep in range(2):
        got, _ = _with_ctx(on, lambda: _operators(N))
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (N, schedule, rep, k, np.max(np.abs(ref[k] - got[k])))
    off.close()
    on.close()


def _work(ctx, fn):
    _with_ctx(ctx, fn)
    return ctx.factor_work()


def test_the_work_counter_shows_what_was_skipped(monkeypatch):
    """f3 = f1 + f2 observed at all three (the north-star model, blocks in the order f1, f2, f3): the (f2, f1) block of the
    factor is zero, 5 / 9 of the dense work remains in the limit of many tiles per block; ordered f3, f1, f2 the (f2, f1)
    block FILLS IN (both rows meet f3's column) and nothing can be skipped; dense noise switches the pattern off."""
    rng = np.random.default_rng(3)
    n = 1536                                  # 12 tiles per block: block boundaries on tile boundaries
    xs = {k: P.ColVecs(np.asfortranarray(rng.standard_normal((2, n)))) for k in ("f1", "f2", "f3")}
    F = P.gppp_sum_model()
    y = rng.standard_normal(3 * n)

    def logpdf(order, noise):
        x = P.BlockData([P.GPPPInput(k, xs[k]) for k in order])
        return lambda: P.logpdf(F(x, noise), y)

    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0, SGP_STRUCT_ZEROS=1)
    e, d = _work(ctx, logpdf(("f1", "f2", "f3"), 0.1))
    T = 36
    assert d == sum(j * (T + 1 - j) for j in range(T))       # the bordered row counts as a tile row
    assert 0.50 * d < e < 0.62 * d, (e, d)
    e2, d2 = _work(ctx, logpdf(("f3", "f1", "f2"), 0.1))
    assert e2 == d2 == d
    B = rng.standard_normal((3 * n, 4))
    e3, d3 = _work(ctx, logpdf(("f1", "f2", "f3"), 0.2 * np.eye(3 * n) + 0.01 * B @ B.T))
    assert e3 == d3
    ctx.close()
    ctx0 = _ctx(monkeypatch, 11, SGP_DATAFLOW=0, SGP_STRUCT_ZEROS=0)
    e4, d4 = _work(ctx0, logpdf(("f1", "f2", "f3"), 0.1))
    assert e4 == d4 == d
    ctx0.close()


@pytest.mark.parametrize("order", [("a", "b", "ab", "c", "ca"), ("ab", "c", "a", "ca", "b"), ("c", "b", "a", "ca", "ab")])
def test_fill_in_and_ragged_blocks_against_the_oracle(monkeypatch, order):
    """Five processes over three independent atoms -- a, b, c, ab = a + b, ca = c + 2 a -- in orders that produce different
    fill-in, block sizes that put the block boundaries inside tiles; logpdf, posterior moments and a draw against the oracle's
    dense computation, and bit-identical with the skipping off, on the launch-based and on the dataflow schedule."""
    rng = np.random.default_rng(11)
    sizes = dict(a=301, b=517, ab=260, c=433, ca=389)

    def build(api):
        gpc = api.GPC()
        a = api.atomic(api.GP(api.SEKernel()), gpc)
        b = api.atomic(api.GP(api.Matern52Kernel()), gpc)
        c = api.atomic(api.GP(api.Matern32Kernel()), gpc)
        return api.GPPP({"a": a, "b": b, "c": c, "ab": a + b, "ca": c + 2.0 * a}, gpc)

    import models
    Fo, Fp = build(models.oracle_api()), build(models.product_api())
    pts = {k: rng.standard_normal(sizes[k]) for k in order}
    xo = ost.BlockData([ost.GPPPInput(k, pts[k]) for k in order])
    xp = P.BlockData([P.GPPPInput(k, pts[k]) for k in order])
    N = sum(sizes.values())
    y = rng.standard_normal(N)
    noise = 0.2 + 0.1 * rng.random(N)
    t = rng.standard_normal(50)
    Z = np.asfortranarray(rng.standard_normal((N, 2)))
    lo = oagp.logpdf(Fo(xo, noise), y)
    mo, vo = oagp.posterior(Fo(xo, noise), y).mean_and_var(ost.GPPPInput("ca", t))

    def run():
        fx = Fp(xp, noise)
        post = P.posterior(fx, y)
        m, v = post.mean_and_var(P.GPPPInput("ca", t))
        return dict(lp=np.array([P.logpdf(fx, y)]), m=np.asarray(m), v=np.asarray(v), r=np.asarray(P.rand(None, fx, 2, Z=Z)))

    outs = {}
    for name, env in (("launch-off", dict(SGP_DATAFLOW=0, SGP_STRUCT_ZEROS=0)), ("launch-on", dict(SGP_DATAFLOW=0, SGP_STRUCT_ZEROS=1)),
                      ("df-on", dict(SGP_DATAFLOW=1, SGP_DF_FAT_MAX_N=0, SGP_STRUCT_ZEROS=1)),
                      ("dffat-on", dict(SGP_DATAFLOW=1, SGP_DF_FAT_MAX_N=1 << 30, SGP_STRUCT_ZEROS=1))):
        ctx = _ctx(monkeypatch, 11, **env)
        outs[name] = _with_ctx(ctx, run)
        if name == "launch-on":
            e, d = ctx.factor_work()
            assert e < d, (order, e, d)
        ctx.close()
    ref = outs["launch-off"]
    assert abs(ref["lp"][0] - lo) <= 1e-10 * abs(lo)
    np.testing.assert_allclose(ref["m"], mo, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(ref["v"], vo, rtol=1e-8, atol=1e-9)
    for name, o in outs.items():
        for k in ref:
            assert np.array_equal(ref[k], o[k]), (order, name, k)


def _symbolic_work(K, n_border_tiles=1, tile=128):
    """Independent statement of the pattern: tile-level non-zeros of the DENSE covariance the oracle computed (exact zeros
    included), symbolic factorisation with fill-in, and the tile products of the contractions -- executed and dense."""
    N = K.shape[0]
    T = -(-N // tile)
    Tr = T + n_border_tiles
    nz = np.zeros((Tr, T), bool)
    for i in range(T):
        for k in range(i + 1):
            nz[i, k] = i == k or bool(np.any(K[i * tile:(i + 1) * tile, k * tile:(k + 1) * tile] != 0.0))
    nz[T:, :] = True
    executed = dense = 0
    for j in range(T):
        for i in range(j, Tr):
            shared = int(np.count_nonzero(nz[i, :j] & nz[j, :j]))
            if shared:
                nz[i, j] = True
            if nz[i, j]:
                executed += shared
            dense += j
    return executed, dense


@pytest.mark.parametrize("order", [("a", "b", "ab", "c", "ca"), ("c", "b", "a", "ca", "ab"), ("b", "c", "a", "ab", "ca")])
def test_the_pattern_is_the_one_the_dense_covariance_implies(monkeypatch, order):
    """The library derives the pattern from the spec's block-pair table; here it is derived from the exact zeros of the
    covariance matrix the ORACLE computed, tile by tile, with its own symbolic factorisation: the work counters must agree
    exactly (ragged block boundaries, fill-in, the bordered row)."""
    import models
    rng = np.random.default_rng(5)
    sizes = dict(a=301, b=517, ab=260, c=433, ca=389)

    def build(api):
        gpc = api.GPC()
        a = api.atomic(api.GP(api.SEKernel()), gpc)
        b = api.atomic(api.GP(api.Matern52Kernel()), gpc)
        c = api.atomic(api.GP(api.Matern32Kernel()), gpc)
        return api.GPPP({"a": a, "b": b, "c": c, "ab": a + b, "ca": c + 2.0 * a}, gpc)

    Fo, Fp = build(models.oracle_api()), build(models.product_api())
    pts = {k: rng.standard_normal(sizes[k]) for k in order}
    Ko = Fo.cov(ost.BlockData([ost.GPPPInput(k, pts[k]) for k in order]))
    want = _symbolic_work(Ko)
    xp = P.BlockData([P.GPPPInput(k, pts[k]) for k in order])
    y = rng.standard_normal(sum(sizes.values()))
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0, SGP_STRUCT_ZEROS=1)
    got = _work(ctx, lambda: P.logpdf(Fp(xp, 0.3), y))
    ctx.close()
    assert got == want, (order, got, want)
    assert want[0] < want[1]


def test_reordering_the_blocks_restores_the_skipping(monkeypatch):
    """stheno.jl_amd/ordering.py: observed in the order (f3, f1, f2) the sum model's factor fills in completely; the order
    fill_reducing_order suggests skips 0.4 of the tile products again, and logpdf / posterior moments agree with the
    caller's order to rounding (they do not depend on the order of the observations)."""
    rng = np.random.default_rng(8)
    n = 1280
    xs = {k: P.ColVecs(np.asfortranarray(rng.standard_normal((2, n)))) for k in ("f1", "f2", "f3")}
    F = P.gppp_sum_model()
    x = P.BlockData([P.GPPPInput(k, xs[k]) for k in ("f3", "f1", "f2")])
    y = rng.standard_normal(3 * n)
    noise = 0.2 + 0.1 * rng.random(3 * n)
    perm = P.fill_reducing_order(F, x)
    x2, (y2,), noise2 = P.permute_blocks(x, perm, y, noise=noise)
    t = P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((2, 20)))))
    ctx = _ctx(monkeypatch, 11, SGP_DATAFLOW=0, SGP_STRUCT_ZEROS=1)
    lp = _with_ctx(ctx, lambda: P.logpdf(F(x, noise), y))
    e, d = ctx.factor_work()
    assert e == d
    lp2 = _with_ctx(ctx, lambda: P.logpdf(F(x2, noise2), y2))
    e2, d2 = ctx.factor_work()
    assert d2 == d and e2 < 0.62 * d2, (e2, d2)
    assert abs(lp - lp2) <= 1e-11 * abs(lp)
    m, v = _with_ctx(ctx, lambda: P.posterior(F(x, noise), y).mean_and_var(t))
    m2, v2 = _with_ctx(ctx, lambda: P.posterior(F(x2, noise2), y2).mean_and_var(t))
    np.testing.assert_allclose(m, m2, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(v, v2, rtol=1e-9, atol=1e-11)
    ctx.close()


def test_the_gradient_skips_the_structural_zeros_and_keeps_its_bits(monkeypatch):
    """Round 5: the gradient's factorisation of [K ; (y - m)' ; I] runs under the pattern as well -- the identity rows carry the
    pattern of inv(L)' (the closure of the factor's; tests/sz_pattern_host.cpp checks it against numerical inverses) -- the
    product C^-1 = inv(L)' inv(L) contracts only the k tiles both operand tiles have and computes only the tiles a block pair
    with terms (or the diagonal) reads.  Value, d/dy, d/d noise (scalar and diagonal), the term gradients and the input-point
    gradients are bit-identical with the skipping off; the work counter shows the skipped share; the oracle's cotangents
    agree."""
    from test_gpu_fused_potrf import _problem
    N = 2900
    F, x, xs, y = _problem(N)
    noise_d = 0.05 + np.random.default_rng(1).random(N)
    res = {}
    for sz in (0, 1):
        ctx = _ctx(monkeypatch, 11, SGP_STRUCT_ZEROS=sz)

        def run():
            g = P.logpdf_and_gradient(F(x, 0.1), y, inputs=True)
            work = ctx.factor_work()
            gd = P.logpdf_and_gradient(F(x, noise_d), y)
            return g, gd, work
        res[sz] = _with_ctx(ctx, run)
        ctx.close()
    (g0, gd0, w0), (g1, gd1, w1) = res[0], res[1]
    assert w0[0] == w0[1] and w1[0] < 0.80 * w1[1], (w0, w1)   # (the identity rows have less to skip than K's)
    assert w1[1] == w0[1]                                     # the dense count of the bordered shape is the same number
    assert g0["logpdf"] == g1["logpdf"] and gd0["logpdf"] == gd1["logpdf"]
    for k in ("y", "mean"):
        assert np.array_equal(g0[k], g1[k]), k
    assert g0["noise"] == g1["noise"] and np.array_equal(gd0["noise"], gd1["noise"])
    for t0, t1 in zip(g0["terms"], g1["terms"]):
        assert t0["d_coef"] == t1["d_coef"] and t0["d_inscale"] == t1["d_inscale"], (t0, t1)
    for a0, a1 in zip(g0["inputs"], g1["inputs"]):
        assert np.array_equal(a0, a1)
    # the oracle: d logpdf / d sigma^2 = tr(G), G = (alpha alpha' - C^-1) / 2, by a plain dense solve
    Fo = ost_sum_model()
    xo = ost.BlockData([ost.GPPPInput(k, oagp.kf.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
    _, C = oagp.mean_and_cov(Fo(xo, 0.1))
    C = np.asarray(C)
    alpha = np.linalg.solve(C, y)
    tr = 0.5 * (alpha @ alpha - np.trace(np.linalg.inv(C)))
    assert abs(g1["noise"] - tr) <= 1e-8 * abs(tr)


def ost_sum_model():
    from oracle import reference_model as orm
    return orm.gppp_sum()


@pytest.mark.parametrize("N", [900, 70000])
def test_the_elbo_skips_the_structural_zeros_of_kzz(monkeypatch, N):
    """Round 5: inducing points spread over the independent processes of a programme give K(z,z) the same exact zero blocks
    K(x,x) has; its factorisation (inside both VFE pipelines: the bordered one and, beyond 65 536 data points, the row-chunked
    one) runs under the pattern.  ELBO and the approximate posterior's moments bit-identical with the skipping off."""
    from test_gpu_fused_potrf import _problem
    F, x, xs, y = _problem(N)
    rng = np.random.default_rng(4)
    z = P.BlockData([P.GPPPInput(k, P.ColVecs(np.asfortranarray(rng.standard_normal((3, m))))) for k, m in (("f1", 300), ("f2", 420), ("f3", 333))])
    xs_new = P.BlockData([P.GPPPInput("f3", P.ColVecs(np.asfortranarray(rng.standard_normal((3, 25)))))])
    res = {}
    for sz in (0, 1):
        ctx = _ctx(monkeypatch, 11, SGP_STRUCT_ZEROS=sz)

        def run():
            fx, vfe = F(x, 0.1), P.VFE(F(z, 1e-6))
            e = P.elbo(vfe, fx, y)
            work = ctx.factor_work()
            m, v = P.posterior(vfe, fx, y).mean_and_var(xs_new)
            return e, work, np.asarray(m), np.asarray(v)
        res[sz] = _with_ctx(ctx, run)
        ctx.close()
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
    assert np.isfinite(res[1][0])
    if N < 5000:
        assert res[1][0] < P.logpdf(F(x, 0.1), y) + 1e-6 * abs(res[1][0])   # ELBO <= logpdf
