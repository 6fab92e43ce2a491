"""The oracle's VFE formulas (oracle/abstractgps.py: elbo, posterior_vfe -- SURVEY Appendix A.6, the arithmetic
/root/reference/src/gp/sparse_finite_gp.jl:52-62 delegates to AbstractGPs; reference pins test/gp/sparse_finite_gp.jl:37-41
are inequalities only) against the DENSE statements of Titsias (2009) in oracle/titsias_dense.py: plain solve / slogdet on
N x N matrices, inducing points that are not data points, isotropic and non-isotropic Sigma_y.  Before round 4 the
factorised A.6 expression was pinned only against a second transcription of itself (and at z = x, where the trace term
vanishes); these tests fail if its DTC part, its trace term or the posterior's two correction terms are wrong."""
import numpy as np
import pytest

import models
import oracle.abstractgps as agp
import oracle.kernelfunctions as kf
import oracle.stheno as st
import oracle.titsias_dense as td
from titsias_cases import gppp_case, single_gp_case

TOL = 1e-10


def _single_oracle(c):
    f = c["coef"] * st.stretch(st.atomic(agp.GP(c["mean"], kf.Matern52Kernel()), st.GPC()), 1.0 / c["ell"])
    return f


@pytest.mark.parametrize("noise_kind", ["scalar", "diag"])
def test_elbo_and_vfe_posterior_single_gp_against_dense_titsias(noise_kind):
    c = single_gp_case(noise_kind)
    # the dense side uses NO oracle code: kernel written out in titsias_dense, prior variance coef^2, constant mean
    s2 = c["coef"] ** 2
    Kff = s2 * td.matern52(c["X"], c["X"], c["ell"])
    Kfu = s2 * td.matern52(c["X"], c["Z"], c["ell"])
    Kuu = s2 * td.matern52(c["Z"], c["Z"], c["ell"]) + c["jitter"] * np.eye(c["Z"].shape[1])
    Ksu = s2 * td.matern52(c["Xs"], c["Z"], c["ell"])
    Kss = s2 * td.matern52(c["Xs"], c["Xs"], c["ell"])
    N, NS = c["X"].shape[1], c["Xs"].shape[1]
    m, ms = np.full(N, c["coef"] * c["mean"]), np.full(NS, c["coef"] * c["mean"])
    want = td.elbo_dense(Kff, Kfu, Kuu, m, c["y"], c["sy"])
    want_mean, want_cov = td.approx_posterior_dense(Kfu, Kuu, m, c["y"], c["sy"], Ksu, Kss, ms)
    f = _single_oracle(c)
    fx, fz = f(kf.ColVecs(c["X"]), c["noise"]), f(kf.ColVecs(c["Z"]), c["jitter"])
    got = agp.elbo(agp.VFE(fz), fx, c["y"])
    assert abs(got - want) <= TOL * abs(want), (got, want)
    assert got < agp.logpdf(fx, c["y"])                                  # the reference's own pin (:40-41)
    post = agp.posterior_vfe(agp.VFE(fz), fx, c["y"])
    xs = kf.ColVecs(c["Xs"])
    assert np.abs(post.mean(xs) - want_mean).max() <= TOL * np.abs(want_mean).max()
    assert np.abs(post.cov(xs) - want_cov).max() <= TOL * np.abs(want_cov).max()
    assert np.abs(post.var(xs) - np.diag(want_cov)).max() <= TOL * np.abs(want_cov).max()


def test_elbo_and_vfe_posterior_across_processes_of_a_gppp_against_dense_titsias():
    c = gppp_case()
    fo, go = models.gppp_docstring(models.oracle_api())
    F = st.GPPP(fo, go)
    x, z, xs = st.GPPPInput("f3", c["x"]), st.GPPPInput("f1", c["z"]), st.GPPPInput("f2", c["xs"])
    # prior blocks from the oracle's recursion (derived_gp.jl:31-60 restated), the bound and posterior dense
    Kff, Kfu, Kuu = F.cov(x), F.cov(x, z), F.cov(z) + c["jitter"] * np.eye(len(c["z"]))
    want = td.elbo_dense(Kff, Kfu, Kuu, F.mean(x), c["y"], c["noise"])
    want_mean, want_cov = td.approx_posterior_dense(Kfu, Kuu, F.mean(x), c["y"], c["noise"], F.cov(xs, z), F.cov(xs), F.mean(xs))
    fx, fz = F(x, c["noise"]), F(z, c["jitter"])
    got = agp.elbo(agp.VFE(fz), fx, c["y"])
    assert abs(got - want) <= TOL * abs(want), (got, want)
    post = agp.posterior_vfe(agp.VFE(fz), fx, c["y"])
    assert np.abs(post.mean(xs) - want_mean).max() <= TOL * max(1.0, np.abs(want_mean).max())
    assert np.abs(post.cov(xs) - want_cov).max() <= TOL * np.abs(want_cov).max()


def test_dense_statement_detects_a_wrong_trace_term():
    """The old pins (elbo == logpdf at z = x, elbo < logpdf) are blind to the trace term; the dense statement is not:
    dropping it moves the value by far more than the tolerance."""
    c = single_gp_case("diag")
    s2 = c["coef"] ** 2
    Kff = s2 * td.matern52(c["X"], c["X"], c["ell"])
    Kfu = s2 * td.matern52(c["X"], c["Z"], c["ell"])
    Kuu = s2 * td.matern52(c["Z"], c["Z"], c["ell"]) + c["jitter"] * np.eye(c["Z"].shape[1])
    N = c["X"].shape[1]
    m = np.full(N, c["coef"] * c["mean"])
    full = td.elbo_dense(Kff, Kfu, Kuu, m, c["y"], c["sy"])
    Qff = Kfu @ np.linalg.solve(Kuu, Kfu.T)
    dtc_only = td.elbo_dense(Qff, Kfu, Kuu, m, c["y"], c["sy"])         # Kff := Qff removes the trace term
    assert dtc_only - full > 1.0


def test_c4_golden_generator_elbo_against_dense_titsias():
    """tests/golden/make_baseline_golden.py: elbo_case is the standalone script that produced the committed c4 golden
    (M = 4096, N = 262144 -- far too large for a dense statement).  The SAME function on a small problem must give the
    dense Titsias value: the c4 golden then certifies GPU = Titsias, not GPU = a transcription of A.6."""
    import importlib.util
    import math
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_baseline_golden", os.path.join(here, "golden", "make_baseline_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rng = np.random.default_rng(5)
    D, N, M, zn = 8, 700, 60, 1e-6
    X, y = rng.standard_normal((D, N)), rng.standard_normal(N)
    got = gen.elbo_case("small", X, y, M, zn, None)["elbo"]
    Xl = X / math.sqrt(D)
    Z = Xl[:, np.random.default_rng(7).permutation(N)[:M]]              # the generator's choice of inducing points
    want = td.elbo_dense(td.se(Xl, Xl, 1.0), td.se(Xl, Z, 1.0), td.se(Z, Z, 1.0) + zn * np.eye(M), np.zeros(N), y,
                         np.full(N, gen.SIGMA2))
    assert abs(got - want) <= TOL * abs(want), (got, want)
