"""CPU baseline of bench.py: the oracle restatement (NumPy/SciPy/OpenBLAS -- NOT Julia) timed on the
host cores of the GPU box, on a bounded sample of the benchmarked workload.

TEST INFRASTRUCTURE: only bench.py's `cpu_baseline` leg and tests import this.

Why a blocked driver: SciPy's bundled OpenBLAS `dpotrf` crashes for N >= 32768 and scales badly with
threads on many-core hosts (33 GFLOP/s with 16 threads on the 256-core box in round 1, where an
8-core box reaches 114), so the factorisation here is a left-looking blocked Cholesky over
<= 4096-wide potrf + trsm + one deep dgemm per block column -- the dgemm is where OpenBLAS threads
scale.  The thread count is swept and the best one used, so the CPU gets its best configuration.
"""
from __future__ import annotations

import math
import os
import time

import numpy as np
import scipy.linalg as sla

from . import abstractgps as agp
from . import kernelfunctions as kf
from . import reference_model as orm
from . import stheno as st

NB = 4096


def cholesky_blocked_inplace(A):
    """lower Cholesky of the lower triangle of the row-major square array A, in place
    (LinearAlgebra.cholesky under AbstractGPs.logpdf [EXT])."""
    N = A.shape[0]
    for k in range(0, N, NB):
        k1 = min(N, k + NB)
        if k > 0:
            A[k:, k:k1] -= A[k:, :k] @ A[k:k1, :k].T
        L11 = sla.cholesky(A[k:k1, k:k1], lower=True, check_finite=False)
        A[k:k1, k:k1] = L11
        if k1 < N:
            A[k1:, k:k1] = sla.solve_triangular(L11, A[k1:, k:k1].T, lower=True, check_finite=False).T
    return A


def forward_solve_blocked(L, b):
    N = L.shape[0]
    z = np.array(b, dtype=float, copy=True)
    for k in range(0, N, NB):
        k1 = min(N, k + NB)
        if k > 0:
            z[k:k1] -= L[k:k1, :k] @ z[:k]
        z[k:k1] = sla.solve_triangular(L[k:k1, k:k1], z[k:k1], lower=True, check_finite=False)
    return z


def sweep_threads(n=12288):
    """-> (fastest thread count, {threads: GFLOP/s}) of the blocked Cholesky at size n.

    n = 12288 (three 4096-wide block columns): the two deep dgemm updates carry 3/4 of the flops, as they do at the
    N = 16384 sample that is then timed -- at n = 6144 (round 2) one 4096 potrf + trsm dominated, the sweep came out flat
    and 8 threads "won" on a 256-core host.  The FASTEST count is taken."""
    from threadpoolctl import threadpool_limits
    ncpu = os.cpu_count() or 1
    cands = sorted({c for c in (8, 32, 64, 128) if c <= ncpu} | ({ncpu} if ncpu < 8 else set()))
    rng = np.random.default_rng(0)
    B = rng.standard_normal((n, n // 8))
    S = B @ B.T + n * np.eye(n)
    rates = {}
    for c in cands:
        with threadpool_limits(limits=c):
            A = S.copy()
            t0 = time.perf_counter()
            cholesky_blocked_inplace(A)
            rates[c] = n ** 3 / 3 / (time.perf_counter() - t0) / 1e9
    best = max(cands, key=lambda c: rates[c])
    return best, rates


def _dense_logpdf_timed(fx, y):
    """one full CPU logpdf: assembly (+ noise) -> blocked Cholesky -> forward solve -> value.
    -> (value, t_assemble, t_cholesky, t_rest)"""
    t0 = time.perf_counter()
    m, Cm = agp.mean_and_cov(fx)
    Cm = np.ascontiguousarray(Cm)
    t1 = time.perf_counter()
    cholesky_blocked_inplace(Cm)
    t2 = time.perf_counter()
    z = forward_solve_blocked(Cm, y - m)
    val = -0.5 * (len(y) * agp.LOG2PI + 2.0 * np.log(np.diagonal(Cm)).sum() + z @ z)
    t3 = time.perf_counter()
    return float(val), t1 - t0, t2 - t1, t3 - t2


def measure(kind, D, N_target, blocks, X, y, sigma2, n_sample, elbo_m=0, elbo_znoise=0.0):
    """Times the oracle on the first n_sample points of the workload's own inputs and scales the
    stages to N_target (assembly + solve ~ N^2, Cholesky ~ N^3; the ELBO ~ N).  -> cpu_baseline dict."""
    from threadpoolctl import threadpool_limits
    threads, rates = sweep_threads(min(12288, max(4096, n_sample)))
    host_cores = os.cpu_count() or 1
    ls = math.sqrt(D)
    n = min(n_sample, N_target)
    with threadpool_limits(limits=threads):
        if kind == "elbo":
            f = st.stretch(st.atomic(agp.GP(kf.SEKernel()), st.GPC()), 1.0 / ls)
            Z = X[:, np.random.default_rng(7).permutation(N_target)[:elbo_m]]
            t0 = time.perf_counter()
            val = agp.elbo(agp.VFE(f(kf.ColVecs(Z), elbo_znoise)), f(kf.ColVecs(X[:, :n]), sigma2), y[:n])
            dt = time.perf_counter() - t0
            t_target = dt * N_target / n
            flops = 2.0 * elbo_m ** 2 * n + 2.0 * elbo_m ** 3 / 3
            return {"value": 1.0 / t_target, "unit": "elbo/s", "cores": int(threads), "threads_used": int(threads),
                    "host_cores": int(host_cores), "kind": "port",
                    "sample": (f"oracle restatement (NumPy/SciPy/OpenBLAS, not Julia): elbo with M={elbo_m} on the first "
                               f"N={n} data points: {dt:.2f} s ({flops / dt / 1e9:.0f} GFLOP/s); scaled x{N_target / n:.0f} "
                               f"(linear in N) to N={N_target}"),
                    "measured_s": dt, "measured_gflops": flops / dt / 1e9, "value_at_sample": float(val),
                    "thread_sweep_cholesky_gflops": rates}
        if kind == "gppp3":
            F = orm.gppp_sum()
            lens = [int(round(b * n / N_target)) for b in blocks]
            lens[-1] = n - sum(lens[:-1])
            cuts = np.concatenate([[0], np.cumsum(blocks)]).astype(int)
            xs = [X[:, cuts[i]:cuts[i] + lens[i]] / ls for i in range(3)]
            ys = np.concatenate([y[cuts[i]:cuts[i] + lens[i]] for i in range(3)])
            fx = F(st.BlockData([st.GPPPInput(k, kf.ColVecs(x)) for k, x in zip(("f1", "f2", "f3"), xs)]), sigma2)
            val, ta, tc, tr = _dense_logpdf_timed(fx, ys)
        else:
            f = orm.single_gp(kind, ls)
            val, ta, tc, tr = _dense_logpdf_timed(f(kf.ColVecs(X[:, :n]), sigma2), y[:n])
    r = N_target / n
    t_target = (ta + tr) * r ** 2 + tc * r ** 3
    gf = n ** 3 / 3 / tc / 1e9
    return {"value": 1.0 / t_target, "unit": "logpdf/s", "cores": int(threads), "threads_used": int(threads),
            "host_cores": int(host_cores), "kind": "port",
            "sample": (f"oracle restatement (NumPy/SciPy/OpenBLAS, not Julia) measured at N={n}, D={D}: assembly {ta:.2f} s, "
                       f"blocked Cholesky {tc:.2f} s ({gf:.0f} GFLOP/s on {threads} threads), solve {tr:.2f} s; scaled to "
                       f"N={N_target} as N^2 (assembly + solve) + N^3 (Cholesky)"),
            "measured_s": ta + tc + tr, "measured_cholesky_gflops": gf, "logpdf_at_sample": val,
            "thread_sweep_cholesky_gflops": rates}
