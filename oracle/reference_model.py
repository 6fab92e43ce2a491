"""Oracle-side constructions of the BASELINE.json configurations.  TEST
INFRASTRUCTURE: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
"""
from __future__ import annotations

import numpy as np

from . import abstractgps as agp
from . import kernelfunctions as kf
from . import stheno as st


def gppp_sum():
    """@gppp let f1 = GP(SEKernel()); f2 = GP(Matern52Kernel()); f3 = f1 + f2 end
    (/root/reference/src/gaussian_process_probabilistic_programme.jl:145-149)."""
    gpc = st.GPC()
    f1 = st.atomic(agp.GP(kf.SEKernel()), gpc)
    f2 = st.atomic(agp.GP(kf.Matern52Kernel()), gpc)
    f3 = f1 + f2
    return st.GPPP({"f1": f1, "f2": f2, "f3": f3}, gpc)


def _blockdata(xs):
    return st.BlockData([st.GPPPInput(k, kf.ColVecs(x)) for k, x in zip(("f1", "f2", "f3"), xs)])


def gppp_sum_logpdf(xs, y, s2):
    f = gppp_sum()
    return agp.logpdf(f(_blockdata(xs), s2), y)


def gppp_sum_posterior(xs, y, s2, Xs_f3):
    f = gppp_sum()
    post = agp.posterior(f(_blockdata(xs), s2), y)
    xs_new = st.GPPPInput("f3", kf.ColVecs(Xs_f3))
    return post.mean(xs_new), post.var(xs_new)


def single_gp(kind, lengthscale):
    """Single GP of BASELINE configs C1/C2/C5: kernel `kind`, lengthscale applied as
    stretch(f, 1 / l)."""
    k = {"se": kf.SEKernel, "matern52": kf.Matern52Kernel, "matern32": kf.Matern32Kernel,
         "matern12": kf.Matern12Kernel}[kind]()
    f = st.atomic(agp.GP(k), st.GPC())
    return st.stretch(f, 1.0 / lengthscale)


def single_gp_logpdf(kind, lengthscale, X, y, s2):
    f = single_gp(kind, lengthscale)
    return agp.logpdf(f(kf.ColVecs(X), s2), y)


def cpu_logpdf_timed(kind, lengthscale, X, y, s2):
    """One full CPU logpdf (assembly + `+ s2 I` + dpotrf + dtrtrs + logdet), staged timings.
    Returns (value, seconds_total, seconds_cholesky)."""
    import time
    import scipy.linalg as sla

    t0 = time.perf_counter()
    f = single_gp(kind, lengthscale)
    fx = f(kf.ColVecs(X), s2)
    m, Cm = agp.mean_and_cov(fx)
    t1 = time.perf_counter()
    L = sla.cholesky(Cm, lower=True, overwrite_a=True, check_finite=False)
    t2 = time.perf_counter()
    z = sla.solve_triangular(L, y - m, lower=True, check_finite=False)
    val = -0.5 * (len(y) * agp.LOG2PI + 2.0 * np.log(np.diag(L)).sum() + z @ z)
    t3 = time.perf_counter()
    return float(val), t3 - t0, t2 - t1
