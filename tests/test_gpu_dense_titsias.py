"""-m gpu: the HIP path's ELBO and VFE posterior (sgp_elbo, sgp_sparse_posterior_* through the host mirror) against the
DENSE statements of Titsias (2009) in oracle/titsias_dense.py -- not against the factorised A.6 transcription the c4
golden and the oracle share.  Reference call site: /root/reference/src/gp/sparse_finite_gp.jl:52-62."""
import numpy as np
import pytest

import models
import oracle.stheno as ost
import oracle.titsias_dense as td
import stheno_jl_amd as P
from titsias_cases import gppp_case, single_gp_case

pytestmark = pytest.mark.gpu
TOL = 1e-9   # (north star: 1e-8 relative; observed ~1e-12)


@pytest.mark.parametrize("noise_kind", ["scalar", "diag"])
def test_hip_elbo_and_vfe_posterior_single_gp_against_dense_titsias(noise_kind):
    c = single_gp_case(noise_kind)
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
    f = c["coef"] * P.stretch(P.atomic(P.GP(c["mean"], P.Matern52Kernel()), P.GPC()), 1.0 / c["ell"])
    fx, fz = f(P.ColVecs(c["X"]), c["noise"]), f(P.ColVecs(c["Z"]), c["jitter"])
    got = P.elbo(P.VFE(fz), fx, c["y"])
    assert abs(got - want) <= TOL * abs(want), (got, want)
    post = P.posterior(P.VFE(fz), fx, c["y"])
    xs = P.ColVecs(c["Xs"])
    mean, var = post.mean_and_var(xs)
    assert np.abs(mean - want_mean).max() <= TOL * np.abs(want_mean).max()
    assert np.abs(var - np.diag(want_cov)).max() <= TOL * np.abs(want_cov).max()
    assert np.abs(post.cov(xs) - want_cov).max() <= TOL * np.abs(want_cov).max()


def test_hip_elbo_and_vfe_posterior_across_processes_of_a_gppp_against_dense_titsias():
    c = gppp_case()
    fo, go = models.gppp_docstring(models.oracle_api())
    Fo = ost.GPPP(fo, go)
    xo, zo, xso = ost.GPPPInput("f3", c["x"]), ost.GPPPInput("f1", c["z"]), ost.GPPPInput("f2", c["xs"])
    Kfu, Kuu = Fo.cov(xo, zo), Fo.cov(zo) + c["jitter"] * np.eye(len(c["z"]))
    want = td.elbo_dense(Fo.cov(xo), Kfu, Kuu, Fo.mean(xo), c["y"], c["noise"])
    want_mean, want_cov = td.approx_posterior_dense(Kfu, Kuu, Fo.mean(xo), c["y"], c["noise"], Fo.cov(xso, zo), Fo.cov(xso),
                                                    Fo.mean(xso))
    fp, gp = models.gppp_docstring(models.product_api())
    F = P.GPPP(fp, gp)
    x, z, xs = P.GPPPInput("f3", c["x"]), P.GPPPInput("f1", c["z"]), P.GPPPInput("f2", c["xs"])
    fx, fz = F(x, c["noise"]), F(z, c["jitter"])
    got = P.elbo(P.VFE(fz), fx, c["y"])
    assert abs(got - want) <= TOL * abs(want), (got, want)
    post = P.posterior(P.VFE(fz), fx, c["y"])
    mean, var = post.mean_and_var(xs)
    assert np.abs(mean - want_mean).max() <= TOL * max(1.0, np.abs(want_mean).max())
    assert np.abs(var - np.diag(want_cov)).max() <= TOL * np.abs(want_cov).max()
    assert np.abs(post.cov(xs) - want_cov).max() <= TOL * np.abs(want_cov).max()
