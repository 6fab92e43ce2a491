"""Dense statements of the Titsias (2009) variational bound and approximate posterior -- TEST INFRASTRUCTURE.

The factorised expressions the product and `oracle/abstractgps.py` implement (the arithmetic of
AbstractGPs.elbo / posterior(::VFE, ...) that /root/reference/src/gp/sparse_finite_gp.jl:52-62 delegates to) were, until
round 4, pinned only against transcriptions of THEMSELVES.  This module states the same quantities the way the paper
does, with nothing but dense `solve` / `slogdet` on explicitly formed N x N matrices -- no Cholesky factor, no
A = Uz^-T Kzx Lambda^-1/2, no matrix-inversion or determinant lemma:

    Qff   = Kfu (Kuu + Sigma_z)^-1 Kuf
    elbo  = log N(y | m, Qff + Sigma_y)  -  1/2 tr(Sigma_y^-1 (Kff - Qff))                      (Titsias 2009, eq. 9)
    Sigma = (Kuu + Sigma_z + Kuf Sigma_y^-1 Kfu)^-1
    mean* = m* + K*u Sigma Kuf Sigma_y^-1 (y - m)                                                (eq. 6 with eq. 10)
    cov*  = K** - K*u (Kuu + Sigma_z)^-1 Ku* + K*u Sigma Ku*

(`Kuu + Sigma_z`: AbstractGPs builds the inducing FiniteGP `fz = f(z, jitter)` and uses cov(fz) wherever the paper
writes Kuu.)  O(N^3) and only usable at N of a few hundred: tests/test_oracle_vs_dense_titsias.py holds the oracle's A.6
against it, tests/test_gpu_dense_titsias.py the HIP path.  Only tests import this module.
"""
import numpy as np

LOG2PI = float(np.log(2.0 * np.pi))


def _sym(a):
    return 0.5 * (a + a.T)


def elbo_dense(Kff, Kfu, Kuu_jit, m, y, sy):
    """Kff N x N (prior covariance of the observed process at x, WITHOUT noise), Kfu N x M, Kuu_jit = K(z, z) + Sigma_z,
    m prior mean at x, sy the diagonal of Sigma_y (length N)."""
    Kff, Kfu, Kuu_jit = (np.asarray(a, dtype=np.float64) for a in (Kff, Kfu, Kuu_jit))
    sy = np.asarray(sy, dtype=np.float64)
    n = len(y)
    Qff = _sym(Kfu @ np.linalg.solve(Kuu_jit, Kfu.T))
    C = Qff + np.diag(sy)
    d = np.asarray(y, dtype=np.float64) - m
    sign, logdet = np.linalg.slogdet(C)
    assert sign > 0
    loglik = -0.5 * (n * LOG2PI + logdet + d @ np.linalg.solve(C, d))
    trace = float(np.sum((np.diag(Kff) - np.diag(Qff)) / sy))
    return float(loglik - 0.5 * trace)


def approx_posterior_dense(Kfu, Kuu_jit, m, y, sy, Ksu, Kss, ms):
    """mean and covariance of the approximate posterior at test points: Ksu = K(x*, z), Kss = K(x*, x*), ms = m(x*)."""
    Kfu, Kuu_jit, Ksu, Kss = (np.asarray(a, dtype=np.float64) for a in (Kfu, Kuu_jit, Ksu, Kss))
    sy = np.asarray(sy, dtype=np.float64)
    S = Kuu_jit + Kfu.T @ (Kfu / sy[:, None])                      # Sigma^-1
    d = (np.asarray(y, dtype=np.float64) - m) / sy
    mean = ms + Ksu @ np.linalg.solve(S, Kfu.T @ d)
    cov = Kss - Ksu @ np.linalg.solve(Kuu_jit, Ksu.T) + Ksu @ np.linalg.solve(S, Ksu.T)
    return mean, _sym(cov)


# ---- kernels written out once more, so that a case can be stated with NO oracle code at all ----------------------------
def sqdist(X, Y):
    """X: D x n, Y: D x m (ColVecs layout) -> n x m squared Euclidean distances, direct form."""
    return ((X[:, :, None] - Y[:, None, :]) ** 2).sum(0)


def matern52(X, Y, ell):
    r = np.sqrt(sqdist(X, Y)) / ell
    return (1.0 + np.sqrt(5.0) * r + 5.0 * r * r / 3.0) * np.exp(-np.sqrt(5.0) * r)


def se(X, Y, ell):
    return np.exp(-0.5 * sqdist(X, Y) / (ell * ell))
