"""Oracle restatement of the AbstractGPs.jl arithmetic the reference inherits
(SURVEY.md 8a rows A1-A6, Appendix A.2-A.6).  NumPy/SciPy fp64 (LAPACK dpotrf/dtrtrs through
OpenBLAS, the BLAS family Julia ships).  TEST INFRASTRUCTURE.

[EXT] AbstractGPs.jl compat "0.4, 0.5" (/root/reference/Project.toml:16), not vendored.
Reference call sites: src/gp/sparse_finite_gp.jl:37,45-62 (mean, marginals, rand, elbo, VFE,
posterior), src/gp/util.jl:12-14 (cov between FiniteGPs); FiniteGP / logpdf / rand / posterior
have no call site in src/ -- every Stheno GP inherits them through `<: AbstractGP`
(src/gp/util.jl:2, src/gaussian_process_probabilistic_programme.jl:13).

Any object with methods mean(x), cov(x), cov(x, x'), var(x), var(x, x') is an "AbstractGP"
here (the internal AbstractGPs API, docs/src/internals.md:8-24).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

from . import kernelfunctions as kf

LOG2PI = float(np.log(2.0 * np.pi))


class PosDefException(Exception):
    def __init__(self, info):
        super().__init__(f"matrix is not positive definite; Cholesky failed (info={info})")
        self.info = info


def cholesky_lower(C):
    """L with C = L L' (the reference keeps U = L' of Symmetric(C); same numbers)."""
    try:
        return sla.cholesky(C, lower=True, check_finite=False)
    except sla.LinAlgError as e:  # mirror PosDefException(info)
        msg = str(e)
        info = -1
        for tok in msg.replace("-th", " ").split():
            if tok.isdigit():
                info = int(tok)
                break
        raise PosDefException(info)


# ---- mean functions (AbstractGPs.ZeroMean / ConstMean / CustomMean) ---------------------------
class ZeroMean:
    def __call__(self, x):
        return np.zeros(kf.n_inputs(x))


class ConstMean:
    def __init__(self, c):
        self.c = float(c)

    def __call__(self, x):
        return np.full(kf.n_inputs(x), self.c)


class CustomMean:
    """GP(g, k): mean g.(x) -- g maps one input (float or D-vector) to a float."""

    def __init__(self, g):
        self.g = g

    def __call__(self, x):
        X = kf.as_matrix(x)
        if isinstance(x, kf.ColVecs):
            return np.array([float(self.g(X[:, i])) for i in range(X.shape[1])])
        return np.array([float(self.g(float(v))) for v in X[0]])


class GP:
    """AbstractGPs.GP(mean, kernel); GP(k) is zero-mean, GP(c::Real, k) constant mean."""

    def __init__(self, *args):
        if len(args) == 1:
            self.mean_f, self.kernel = ZeroMean(), args[0]
        else:
            m, self.kernel = args
            if isinstance(m, (int, float)):
                self.mean_f = ConstMean(m)
            elif isinstance(m, (ZeroMean, ConstMean, CustomMean)):
                self.mean_f = m
            else:
                self.mean_f = CustomMean(m)

    def mean(self, x):
        return self.mean_f(x)

    def cov(self, x, x2=None):
        return kf.kernelmatrix(self.kernel, x, x2)

    def var(self, x, x2=None):
        # src/gp/util.jl:5-7 pirates var(::GP, x, x') = kernelmatrix_diag(k, x, x')
        return kf.kernelmatrix_diag(self.kernel, x, x2)

    def __call__(self, x, noise=1e-18):
        return FiniteGP(self, x, noise)


# ---- FiniteGP (A.2) ---------------------------------------------------------------------------
def noise_matrix(noise, n):
    a = np.asarray(noise, dtype=np.float64)
    if a.ndim == 0:
        return float(a) * np.eye(n)
    if a.ndim == 1:
        return np.diag(a)
    return a


def noise_diag(noise, n):
    a = np.asarray(noise, dtype=np.float64)
    if a.ndim == 0:
        return np.full(n, float(a))
    if a.ndim == 1:
        return a
    return np.diag(a)


class FiniteGP:
    """f(x, Sigma_y): Sigma_y scalar -> s2 I, vector -> Diagonal, matrix -> dense; default 1e-18."""

    def __init__(self, f, x, noise=1e-18):
        self.f, self.x, self.noise = f, x, noise

    def __len__(self):
        return kf.n_inputs(self.x) if not hasattr(self.x, "__len__") else len(self.x)


def mean(fx):
    return fx.f.mean(fx.x)


def cov(fx, gx=None):
    if gx is None:
        n = len(fx)
        return fx.f.cov(fx.x) + noise_matrix(fx.noise, n)
    # src/gp/util.jl:12-14: cov(fx, gx) = cov(fx.f, gx.f, fx.x, gx.x), no noise term
    return fx.f.cov_cross(gx.f, fx.x, gx.x)


def var(fx):
    n = len(fx)
    return fx.f.var(fx.x) + noise_diag(fx.noise, n)


def mean_and_cov(fx):
    return mean(fx), cov(fx)


def marginals(fx):
    """(mean, std) of the independent Normal marginals (test/gp/util.jl:15-20)."""
    return mean(fx), np.sqrt(var(fx))


def logpdf(fx, y):
    """-(N log 2pi + logdet C + |U^-T (y - m)|^2) / 2; matrix Y -> one value per column (A.3)."""
    m, Cm = mean_and_cov(fx)
    L = cholesky_lower(Cm)
    Y = np.asarray(y, dtype=np.float64)
    vec = Y.ndim == 1
    if vec:
        Y = Y[:, None]
    Z = sla.solve_triangular(L, Y - m[:, None], lower=True, check_finite=False)
    n = len(m)
    out = -0.5 * (n * LOG2PI + 2.0 * np.log(np.diag(L)).sum() + (Z * Z).sum(0))
    return float(out[0]) if vec else out


def rand(fx, Z):
    """m .+ U' Z with Z = randn(rng, N, S) supplied by the caller (A.4)."""
    m, Cm = mean_and_cov(fx)
    L = cholesky_lower(Cm)
    Z = np.asarray(Z, dtype=np.float64)
    if Z.ndim == 1:
        return m + L @ Z
    return m[:, None] + L @ Z


# ---- exact posterior (A.5) ------------------------------------------------------------------
class PosteriorGP:
    def __init__(self, prior, x, alpha, L, delta):
        self.prior, self.x, self.alpha, self.L, self.delta = prior, x, alpha, L, delta

    def _kxs(self, xs):
        # K(x, x*) as the transpose of the prior's cov(x*, x) (a Stheno cross-covariance for GPPPs)
        return self.prior.cov(xs, self.x).T

    def mean(self, xs):
        return self.prior.mean(xs) + self.prior.cov(xs, self.x) @ self.alpha

    def cov(self, xs, zs=None):
        V = sla.solve_triangular(self.L, self._kxs(xs), lower=True, check_finite=False)
        if zs is None:
            return self.prior.cov(xs) - V.T @ V
        W = sla.solve_triangular(self.L, self._kxs(zs), lower=True, check_finite=False)
        return self.prior.cov(xs, zs) - V.T @ W

    def var(self, xs):
        V = sla.solve_triangular(self.L, self._kxs(xs), lower=True, check_finite=False)
        return self.prior.var(xs) - (V * V).sum(0)

    def mean_and_var(self, xs):
        return self.mean(xs), self.var(xs)

    def __call__(self, xs, noise=1e-18):
        return FiniteGP(self, xs, noise)


def posterior(fx, y):
    m, Cm = mean_and_cov(fx)
    L = cholesky_lower(Cm)
    delta = np.asarray(y, dtype=np.float64) - m
    alpha = sla.cho_solve((L, True), delta, check_finite=False)
    return PosteriorGP(fx.f, fx.x, alpha, L, delta)


# ---- VFE / ELBO (A.6) -----------------------------------------------------------------------
class VFE:
    def __init__(self, fz):
        self.fz = fz


def _vfe_parts(vfe, fx, y):
    fz = vfe.fz
    assert fz.f is fx.f, "VFE requires fz.f === fx.f"
    n = len(fx)
    sy = noise_diag(fx.noise, n)
    a = np.asarray(fx.noise, dtype=np.float64)
    assert a.ndim < 2, "elbo needs isotropic / diagonal observation noise"
    Lz = cholesky_lower(cov(fz))
    Kzx = fx.f.cov(fz.x, fx.x)                      # M x N
    A = sla.solve_triangular(Lz, Kzx, lower=True, check_finite=False) / np.sqrt(sy)[None, :]
    Le = cholesky_lower(A @ A.T + np.eye(A.shape[0]))
    delta = (np.asarray(y, dtype=np.float64) - mean(fx)) / np.sqrt(sy)
    return Lz, A, Le, delta, sy


def elbo(vfe, fx, y):
    Lz, A, Le, delta, sy = _vfe_parts(vfe, fx, y)
    n = len(delta)
    b = sla.solve_triangular(Le, A @ delta, lower=True, check_finite=False)
    tmp = np.log(sy).sum() + 2.0 * np.log(np.diag(Le)).sum() + delta @ delta - b @ b
    dtc = -0.5 * (n * LOG2PI + tmp)
    return float(dtc - 0.5 * ((fx.f.var(fx.x) / sy).sum() - (A * A).sum()))


class ApproxPosteriorGP:
    def __init__(self, prior, z, alpha, Lz, Le):
        self.prior, self.z, self.alpha, self.Lz, self.Le = prior, z, alpha, Lz, Le

    def mean(self, xs):
        return self.prior.mean(xs) + self.prior.cov(xs, self.z) @ self.alpha

    def _b(self, xs):
        B = sla.solve_triangular(self.Lz, self.prior.cov(xs, self.z).T, lower=True, check_finite=False)
        Cb = sla.solve_triangular(self.Le, B, lower=True, check_finite=False)
        return B, Cb

    def var(self, xs):
        B, Cb = self._b(xs)
        return self.prior.var(xs) - (B * B).sum(0) + (Cb * Cb).sum(0)

    def cov(self, xs):
        B, Cb = self._b(xs)
        return self.prior.cov(xs) - B.T @ B + Cb.T @ Cb

    def mean_and_var(self, xs):
        return self.mean(xs), self.var(xs)

    def __call__(self, xs, noise=1e-18):
        return FiniteGP(self, xs, noise)


def posterior_vfe(vfe, fx, y):
    Lz, A, Le, delta, _ = _vfe_parts(vfe, fx, y)
    m_eps = sla.cho_solve((Le, True), A @ delta, check_finite=False)
    alpha = sla.solve_triangular(Lz, m_eps, lower=True, trans="T", check_finite=False)
    return ApproxPosteriorGP(fx.f, vfe.fz.x, alpha, Lz, Le)


# ---- gradient of logpdf (what Zygote derives through the reference:
# examples/getting_started/script.jl:154-213) ---------------------------------------------------
def logpdf_gradient_wrt_cov(fx, y):
    """(logpdf, alpha, G) with G = d logpdf / d C = (alpha alpha' - C^-1) / 2, alpha = C^-1 (y - m).
    d logpdf / d y = -alpha, d logpdf / d m = +alpha, d logpdf / d theta = sum_ij G_ij dC_ij/dtheta."""
    m, Cm = mean_and_cov(fx)
    L = cholesky_lower(Cm)
    delta = np.asarray(y, dtype=np.float64) - m
    alpha = sla.cho_solve((L, True), delta, check_finite=False)
    Cinv = sla.cho_solve((L, True), np.eye(len(m)), check_finite=False)
    z = sla.solve_triangular(L, delta, lower=True, check_finite=False)
    lp = -0.5 * (len(m) * LOG2PI + 2.0 * np.log(np.diag(L)).sum() + z @ z)
    return float(lp), alpha, 0.5 * (np.outer(alpha, alpha) - Cinv)


def elbo_gradient_wrt_cov(vfe, fx, y):
    """elbo (A.6) and its reverse-mode gradient w.r.t. the matrices it is built from -- what Zygote
    derives through AbstractGPs.elbo on the reference path.  With
    Lambda = diag(sy)^-1/2, A = Lz^-1 Kzx Lambda, B = A A' + I, delta = Lambda (y - m),
    u = B^-1 A delta, J = Lz^-T:
        dA     = (I - B^-1 - u u') A + u delta'
        dKxz   = Lambda dA' Lz^-1                                (N x M)
        dKzz   = -1/2 J (B + B^-1 - 2 I + u u') J'               (M x M, w.r.t. Kzz + Sigma_z)
        ddelta = -delta + A' u ;  dy = Lambda ddelta = -dmean
        dvar   = -1/(2 sy)
        dsy    = -1/(2 sy) + var/(2 sy^2) - (ddelta delta + diag(A' dA)) / (2 sy)
    Returns dict(elbo, Kzz, Kxz, var, noise, y, mean); noise is summed for scalar noise."""
    Lz, A, Le, delta, sy = _vfe_parts(vfe, fx, y)
    M, N = A.shape
    I = np.eye(M)
    B = A @ A.T + I
    c = A @ delta
    u = sla.cho_solve((Le, True), c, check_finite=False)
    Binv = sla.cho_solve((Le, True), I, check_finite=False)
    Z = I - Binv - np.outer(u, u)
    S = B + Binv - 2.0 * I + np.outer(u, u)
    J = sla.solve_triangular(Lz, I, lower=True, check_finite=False).T       # Lz^-T
    rsig = 1.0 / np.sqrt(sy)
    dA_T = A.T @ Z + np.outer(delta, u)                                       # N x M
    dKxz = rsig[:, None] * (dA_T @ J.T)
    dKzz = -0.5 * J @ S @ J.T
    ddelta = -delta + A.T @ u
    dy = ddelta * rsig
    v = fx.f.var(fx.x)
    diagdot = (A.T * dA_T).sum(1)
    dsy = -0.5 / sy + 0.5 * v / sy ** 2 - 0.5 * (ddelta * delta + diagdot) / sy
    scalar_noise = np.asarray(fx.noise).ndim == 0
    return dict(elbo=elbo(vfe, fx, y), Kzz=dKzz, Kxz=dKxz, var=-0.5 / sy,
                noise=float(dsy.sum()) if scalar_noise else dsy, y=dy, mean=-dy)
