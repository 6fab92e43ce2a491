"""CPU baseline of bench.py: the oracle restatement (NumPy / SciPy / OpenBLAS / MKL -- NOT Julia) timed on the host cores of
the GPU box, on a bounded sample of the benchmarked workload.

TEST INFRASTRUCTURE: only bench.py's `cpu_baseline` leg and tests import this.

Round 5 (verdict: "a CPU baseline that is the CPU's best"): the leg runs in a SUBPROCESS of its own
(`python -m oracle.cpu_baseline <json>`), so that
  * the threading environment is set BEFORE NumPy / torch load their BLAS: OMP_PLACES=cores, OMP_PROC_BIND=spread and one
    thread per PHYSICAL core at most (thread_siblings_list), and the GPU process's own threads are out of the way;
  * three Cholesky implementations compete at every thread count of a sweep and the FASTEST (implementation, threads) pair
    is the one the sample is timed with -- and named in the record:
        "mkl"      torch.linalg.cholesky on the CPU (the torch wheel ships Intel MKL's LAPACK)
        "dpotrf"   LAPACK dpotrf through SciPy on a Fortran-ordered array (SciPy's bundled OpenBLAS; crashes for N >= 32768,
                   so it is only ever run on the sample)
        "blocked"  left-looking blocked driver over <= 4096-wide potrf + trsm + ONE deep dgemm per block column (the dgemm is
                   where OpenBLAS threads scale; round 1-4's only candidate)
  * the covariance is assembled by rows in a thread pool (NumPy ufuncs release the GIL) with the pairwise distances from
    the GEMM identity |a|^2 + |b|^2 - 2 a'b -- what Distances.jl's `pairwise` does on the reference path
    -- instead of one N x N temporary per arithmetic step (7.5 s at N = 16384 in round 4).  The fast assembly is checked
    against the oracle's own mean_and_cov (the line-by-line restatement) on 384 points of the same inputs before anything is
    timed; a mismatch aborts the leg.
Still a restatement, still "not Julia": a reported baseline, never credit.
"""
from __future__ import annotations

import json
import math
import os
import subprocess
import sys
import time

NB = 4096
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------------------------------------------------
# blocked driver (also imported by tests)
# ----------------------------------------------------------------------------------------------------------------------
def cholesky_blocked_inplace(A):
    """lower Cholesky of the lower triangle of the row-major square array A, in place
    (LinearAlgebra.cholesky under AbstractGPs.logpdf [EXT])."""
    import scipy.linalg as sla
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
    import numpy as np
    import scipy.linalg as sla
    N = L.shape[0]
    z = np.array(b, dtype=float, copy=True)
    for k in range(0, N, NB):
        k1 = min(N, k + NB)
        if k > 0:
            z[k:k1] -= L[k:k1, :k] @ z[:k]
        z[k:k1] = sla.solve_triangular(L[k:k1, k:k1], z[k:k1], lower=True, check_finite=False)
    return z


def physical_cores():
    """one logical CPU per physical core of the CPUs this process may run on (thread_siblings_list), sorted"""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        allowed = list(range(os.cpu_count() or 1))
    seen, firsts = set(), []
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            firsts.append(c)
    return firsts or allowed


# ----------------------------------------------------------------------------------------------------------------------
# the child: everything below runs inside `python -m oracle.cpu_baseline`
# ----------------------------------------------------------------------------------------------------------------------
def _kappa(kind, d2):
    import numpy as np
    if kind == "se":
        return np.exp(-0.5 * d2)                       # KernelFunctions SEKernel: exp(-d^2 / 2)
    d = np.sqrt(d2)
    s5 = math.sqrt(5.0)
    return (1.0 + s5 * d + (5.0 / 3.0) * d2) * np.exp(-s5 * d)   # Matern52Kernel


def _assemble_rows(parts, n, pool_threads):
    """Lower triangle (and more) of the n x n covariance, row block by row block in a thread pool.
    parts(i0, i1) -> the rows i0:i1 of the matrix up to column i1 (ndarray (i1 - i0) x i1)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    A = np.zeros((n, n))
    step = 1024
    blocks = [(i0, min(n, i0 + step)) for i0 in range(0, n, step)]

    def work(b):
        i0, i1 = b
        A[i0:i1, :i1] = parts(i0, i1)
    with ThreadPoolExecutor(max_workers=max(1, pool_threads)) as ex:
        list(ex.map(work, blocks))
    return A


def _pair(kind, Xa, Xb, sqa, sqb):
    import numpy as np
    d2 = sqa[:, None] + sqb[None, :] - 2.0 * (Xa.T @ Xb)          # Distances.jl pairwise(SqEuclidean): GEMM identity ...
    np.maximum(d2, 0.0, out=d2)                                   # ... clamped at zero
    return _kappa(kind, d2)


def fast_cov(kind, xs, sigma2, pool_threads):
    """K + sigma2 I (lower triangle valid) of the benchmarked models; xs: list of D x n_b input blocks (already divided by
    the lengthscale).  kind "se" / "matern52": one block.  "gppp3": blocks (f1, f2, f3), f3 = f1 + f2, f1 ~ SE, f2 ~ Matern-5/2:
      cov(f1, f1) = SE, cov(f2, f2) = M52, cov(f2, f1) = 0, cov(f3, f1) = SE, cov(f3, f2) = M52, cov(f3, f3) = SE + M52
    (/root/reference/src/affine_transformations/addition.jl:26-54, src/gp/atomic_gp.jl:36-41)."""
    import numpy as np
    X = np.concatenate(xs, axis=1)
    n = X.shape[1]
    sq = np.einsum("ij,ij->j", X, X)
    if kind != "gppp3":
        def parts(i0, i1):
            return _pair(kind, X[:, i0:i1], X[:, :i1], sq[i0:i1], sq[:i1])
    else:
        cuts = np.concatenate([[0], np.cumsum([x.shape[1] for x in xs])]).astype(int)
        blk = np.zeros(n, dtype=np.int8)
        for b in range(3):
            blk[cuts[b]:cuts[b + 1]] = b
        uses_se = np.array([True, False, True])       # which atoms a block's process is built from
        uses_m5 = np.array([False, True, True])

        def parts(i0, i1):
            out = np.zeros((i1 - i0, i1))
            rb, cb = blk[i0:i1], blk[:i1]
            for atom_kind, uses in (("se", uses_se), ("matern52", uses_m5)):
                r, c = np.nonzero(uses[rb])[0], np.nonzero(uses[cb])[0]
                if len(r) and len(c):
                    out[np.ix_(r, c)] += _pair(atom_kind, X[:, i0 + r], X[:, c], sq[i0 + r], sq[c])
            return out
    A = _assemble_rows(parts, n, pool_threads)
    A[np.diag_indices(n)] += sigma2
    return A


def _chol(impl, A):
    """lower factor of the (lower triangle of the) row-major array A -> ndarray whose lower triangle is L"""
    import numpy as np
    if impl == "mkl":
        import torch
        L, info = torch.linalg.cholesky_ex(torch.from_numpy(A), upper=False)
        if int(info) != 0:
            raise np.linalg.LinAlgError("not positive definite")
        return L.numpy()
    if impl == "dpotrf":
        import scipy.linalg as sla
        # row-major lower triangle == column-major upper triangle of the transpose: factor A' = U'U in place, L = U'
        c, info = sla.lapack.dpotrf(A.T, lower=0, overwrite_a=1, clean=0)
        if info != 0:
            raise np.linalg.LinAlgError("not positive definite")
        return c.T
    return cholesky_blocked_inplace(A)


def _set_threads(c):
    from threadpoolctl import threadpool_limits
    try:
        import torch
        torch.set_num_threads(int(c))
    except Exception:  # noqa: BLE001
        pass
    return threadpool_limits(limits=int(c))


def _sweep(n, cands, impls):
    """{impl: {threads: GFLOP/s}} of the Cholesky at size n; the fastest pair"""
    import numpy as np
    rng = np.random.default_rng(0)
    B = rng.standard_normal((n, n // 8))
    S = B @ B.T + n * np.eye(n)
    rates, best = {}, (None, None, 0.0)
    for impl in impls:
        rates[impl] = {}
        for c in cands:
            with _set_threads(c):
                A = S.copy()
                t0 = time.perf_counter()
                try:
                    _chol(impl, A)
                except Exception as e:  # noqa: BLE001 -- an implementation that fails is out of the race
                    rates[impl][c] = None
                    rates[impl]["error"] = str(e)[:120]
                    continue
                r = n ** 3 / 3 / (time.perf_counter() - t0) / 1e9
            rates[impl][c] = r
            if r > best[2]:
                best = (impl, c, r)
    return rates, best


def _child(a):
    import numpy as np
    import scipy.linalg as sla
    from . import abstractgps as agp
    from . import kernelfunctions as kf
    from . import reference_model as orm
    from . import stheno as st
    kind, D, N_target, blocks, sigma2, n_sample = a["kind"], a["D"], a["N_target"], a["blocks"], a["sigma2"], a["n_sample"]
    rng = np.random.default_rng(123456)                  # bench_configs.make_inputs
    X = np.asfortranarray(rng.standard_normal((D, N_target)))
    y = rng.standard_normal(N_target)
    phys = physical_cores()
    ncpu = len(phys)
    cands = sorted({c for c in (8, 16, 32, 64, 128, 256) if c <= ncpu} | {ncpu})
    impls = ["mkl", "dpotrf", "blocked"]
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001
        impls.remove("mkl")
    ls = math.sqrt(D)
    n = min(n_sample, N_target)
    sweep_n = int(min(8192, max(4096, n)))
    rates, (impl, threads, _) = _sweep(sweep_n, cands, impls)
    env_note = {k: os.environ.get(k) for k in ("OMP_PLACES", "OMP_PROC_BIND", "OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS")}
    base = {"cores": int(threads), "threads_used": int(threads), "host_cores": int(os.cpu_count() or 1),
            "physical_cores_visible": int(ncpu), "kind": "port", "cholesky_impl": impl,
            "cholesky_sweep_gflops": rates, "cholesky_sweep_n": sweep_n, "thread_env": env_note}
    with _set_threads(threads):
        if kind == "elbo":
            M, zn = a["elbo_m"], a["elbo_znoise"]
            f = st.stretch(st.atomic(agp.GP(kf.SEKernel()), st.GPC()), 1.0 / ls)
            Z = X[:, np.random.default_rng(7).permutation(N_target)[:M]]
            t0 = time.perf_counter()
            val = agp.elbo(agp.VFE(f(kf.ColVecs(Z), zn)), f(kf.ColVecs(X[:, :n]), sigma2), y[:n])
            dt = time.perf_counter() - t0
            t_target = dt * N_target / n
            flops = 2.0 * M ** 2 * n + 2.0 * M ** 3 / 3
            base.update({"value": 1.0 / t_target, "unit": "elbo/s",
                         "sample": (f"oracle restatement (NumPy/SciPy/OpenBLAS, not Julia): elbo with M={M} on the first N={n} data "
                                    f"points: {dt:.2f} s ({flops / dt / 1e9:.0f} GFLOP/s on {threads} threads); scaled x{N_target / n:.0f} "
                                    f"(linear in N) to N={N_target}"),
                         "measured_s": dt, "measured_gflops": flops / dt / 1e9, "value_at_sample": float(val)})
            return base
        if kind == "gppp3":
            lens = [int(round(b * n / N_target)) for b in blocks]
            lens[-1] = n - sum(lens[:-1])
            cuts = np.concatenate([[0], np.cumsum(blocks)]).astype(int)
            xs = [np.ascontiguousarray(X[:, cuts[i]:cuts[i] + lens[i]] / ls) for i in range(3)]
            ys = np.concatenate([y[cuts[i]:cuts[i] + lens[i]] for i in range(3)])

            def oracle_fx(k):
                F = orm.gppp_sum()
                return F(st.BlockData([st.GPPPInput(nm, kf.ColVecs(x[:, :k])) for nm, x in zip(("f1", "f2", "f3"), xs)]), sigma2)
            small = [x[:, :128] for x in xs]
        else:
            xs = [np.ascontiguousarray(X[:, :n] / ls)]
            ys = y[:n]

            def oracle_fx(k):
                return orm.single_gp(kind, ls)(kf.ColVecs(X[:, :k]), sigma2)
            small = [xs[0][:, :384]]
        # the fast assembly against the line-by-line restatement, on a small sample of the same inputs
        _, C_or = agp.mean_and_cov(oracle_fx(small[0].shape[1]))
        C_fast = fast_cov(kind, small, sigma2, 4)
        err = float(np.max(np.abs(np.tril(C_fast) - np.tril(np.asarray(C_or)))))
        if not err <= 1e-12:
            raise SystemExit(f"cpu_baseline: fast assembly disagrees with the oracle's mean_and_cov by {err:g}")
        pool = min(32, threads)
        # The fastest (implementation, threads) pair of the sweep first; a pair that FAILS on the real matrix (round 6: a threaded
        # LAPACK reported "not positive definite" for this well-conditioned matrix at N = 32768 on one 256-core host) is recorded
        # and the next-fastest pair takes over -- the matrix is assembled again, the factorisations work in place.
        ranked = sorted(((r, i, c) for i in impls for c, r in rates.get(i, {}).items() if isinstance(c, int) and r), reverse=True)
        order = [(impl, threads)] + [(i, c) for _, i, c in ranked if (i, c) != (impl, threads)]
        failures, Lm = [], None
        for impl, threads in order[:4]:
            t0 = time.perf_counter()
            with _set_threads(1):                       # the row blocks are the parallelism; BLAS stays single-threaded inside
                Cm = fast_cov(kind, xs, sigma2, pool)
            t1 = time.perf_counter()
            try:
                with _set_threads(threads):
                    Lm = _chol(impl, Cm)
                if not np.all(np.isfinite(np.diagonal(Lm))) or np.min(np.diagonal(Lm)) <= 0.0:
                    raise np.linalg.LinAlgError("non-finite or non-positive diagonal in the factor")
            except np.linalg.LinAlgError as e:
                failures.append(f"{impl} on {threads} threads: {e}")
                Lm = None
                continue
            break
        if Lm is None:
            raise SystemExit("cpu_baseline: every Cholesky implementation failed on the sample matrix: " + "; ".join(failures))
        base.update({"cores": int(threads), "threads_used": int(threads), "cholesky_impl": impl})
        if failures:
            base["cholesky_failures_on_this_host"] = failures
        t2 = time.perf_counter()
        z = forward_solve_blocked(Lm, ys)
        val = -0.5 * (len(ys) * agp.LOG2PI + 2.0 * np.log(np.diagonal(Lm)).sum() + z @ z)
        t3 = time.perf_counter()
    ta, tc, tr = t1 - t0, t2 - t1, t3 - t2
    r = N_target / n
    t_target = (ta + tr) * r ** 2 + tc * r ** 3
    gf = n ** 3 / 3 / tc / 1e9
    what = {"mkl": "torch.linalg.cholesky (MKL LAPACK)", "dpotrf": "LAPACK dpotrf (SciPy's OpenBLAS)",
            "blocked": "blocked potrf/trsm/dgemm driver (OpenBLAS)"}[impl]
    base.update({"value": 1.0 / t_target, "unit": "logpdf/s",
                 "sample": (f"CPU restatement (NumPy/SciPy/MKL/OpenBLAS, not Julia) measured at N={n}, D={D}: row-blocked assembly {ta:.2f} s "
                            f"({pool} threads), Cholesky {tc:.2f} s = {gf:.0f} GFLOP/s with {what} on {threads} threads (fastest of "
                            f"{len(impls)} implementations x {len(cands)} thread counts, threads bound to physical cores), solve {tr:.2f} s; "
                            f"scaled to N={N_target} as N^2 (assembly + solve) + N^3 (Cholesky)"),
                 "measured_s": ta + tc + tr, "measured_cholesky_gflops": gf, "logpdf_at_sample": float(val),
                 "assembly_check_max_abs_err_vs_oracle": err})
    return base


def measure(kind, D, N_target, blocks, X, y, sigma2, n_sample, elbo_m=0, elbo_znoise=0.0):
    """Times the restatement on the first n_sample points of the workload's own inputs (regenerated in the child from the
    same seed) and scales the stages to N_target (assembly + solve ~ N^2, Cholesky ~ N^3; the ELBO ~ N) -> cpu_baseline dict."""
    phys = physical_cores()
    env = dict(os.environ)
    env.update({"OMP_PLACES": "cores", "OMP_PROC_BIND": "spread", "OMP_NUM_THREADS": str(len(phys)),
                "MKL_NUM_THREADS": str(len(phys)), "OPENBLAS_NUM_THREADS": str(min(len(phys), 128)),
                "MKL_DYNAMIC": "FALSE", "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
    env.pop("HIP_VISIBLE_DEVICES", None)
    env["CUDA_VISIBLE_DEVICES"] = ""                    # the child never touches the GPU
    args = {"kind": kind, "D": int(D), "N_target": int(N_target), "blocks": None if blocks is None else [int(b) for b in blocks],
            "sigma2": float(sigma2), "n_sample": int(n_sample), "elbo_m": int(elbo_m), "elbo_znoise": float(elbo_znoise)}
    p = subprocess.run([sys.executable, "-m", "oracle.cpu_baseline", json.dumps(args)], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    if p.returncode != 0:
        raise RuntimeError("cpu baseline child failed: " + (p.stderr or p.stdout)[-2000:])
    return json.loads(p.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    # the child: pin to one logical CPU per physical core before any BLAS spins its threads up
    try:
        os.sched_setaffinity(0, set(physical_cores()))
    except (AttributeError, OSError):  # pragma: no cover
        pass
    print(json.dumps(_child(json.loads(sys.argv[1]))))
