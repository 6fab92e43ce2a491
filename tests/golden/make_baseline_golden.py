"""Generates tests/golden/baseline_configs.json: CPU known-answer values for the BASELINE.json
configurations at their FULL sizes (c1..c5, n4k, and the north-star target model), so that the
`-m gpu` suite and bench.py can pin the HIP path where it is actually benchmarked.

Standalone on purpose: NumPy/SciPy only, imports nothing from this repository (neither the
product nor oracle/), so it is an independent statement of the arithmetic of SURVEY.md
Appendix A (AbstractGPs logpdf / posterior / elbo, KernelFunctions SE / Matern-5/2) applied to
the models of SURVEY.md 8d:

  c1     single GP, SE,         N = 2048,   D = 2
  n4k    single GP, Matern-5/2, N = 4096,   D = 8
  c2     single GP, SE,         N = 16384,  D = 8
  c3     @gppp f3 = f1 + f2 (f1 SE, f2 Matern-5/2; /root/reference/src/
         gaussian_process_probabilistic_programme.jl:145-149), BlockData blocks
         (f1, 10923), (f2, 10923), (f3, 10922), total N = 32768, D = 4
  c4     sparse ELBO (VFE), SE, M = 4096 inducing points, N = 262144, D = 8, Sigma_z = 1e-6 I
  c5     single GP, Matern-5/2, N = 65536,  D = 8
  n32k   single GP, Matern-5/2, N = 32768,  D = 8   (round 3: the largest size the dataflow factorisation serves)
  target c3's model at N = 65536, D = 8, blocks 21846 / 21845 / 21845 (the north-star run)
  w4k    every input transformation of /root/reference/src/affine_transformations/compose.jl in one programme
         (round 3; the cases above only stretch and add), N = 4096, D = 3, a ~ GP(SE), b ~ GP(Matern-5/2):
           g1 = select(stretch(a, 1 / l), [1, 2])        g1(x) = a(x[1:2] / l),  l = sqrt(2)
           g2 = select(periodic(b, 0.3), 3)              g2(x) = b([cos(2 pi 0.3 x[3]), sin(2 pi 0.3 x[3])])
           g3 = shift(g1, [0.4, -0.2, 0.1])              g3(x) = g1(x - s)
           h  = g1 + 2 g2 - 0.5 g3
         logpdf(h(X, 0.1), y) and the posterior of g1 at 64 points, stated point-wise below (WarpModel)

Inputs (identical to bench.py `make_inputs`): rng = default_rng(123456);
X = rng.standard_normal((D, N)); y = rng.standard_normal(N); lengthscale sqrt(D) (inputs divided
by sqrt(D)); sigma^2 = 0.1 (as /root/reference/test/gp/util.jl:82); zero mean.  Posterior
mean / var are recorded at 64 points Xs = default_rng(987).standard_normal((D, 64)) (for the
@gppp models: of process f3).

SciPy's bundled OpenBLAS dpotrf crashes for N >= 32768, so the factorisation is a blocked
left-looking Cholesky over <= 4096-wide potrf + trsm + gemm.  Only the lower triangle is ever
touched.  Peak memory ~36 GB for the two N = 65536 configurations.

  python tests/golden/make_baseline_golden.py            # all configs (~40 min on 8 cores)
  python tests/golden/make_baseline_golden.py c1 c2      # a subset (merged into the JSON)
"""
import json
import math
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "baseline_configs.json")
LOG2PI = math.log(2.0 * math.pi)
SIGMA2 = 0.1
NB = 4096          # Cholesky block
TB = 2048          # assembly tile
NS = 64            # posterior points


def inputs(N, D):
    rng = np.random.default_rng(123456)
    X = rng.standard_normal((D, N))
    y = rng.standard_normal(N)
    return X, y


def kappa(kind, d2):
    if kind == "se":
        return np.exp(-0.5 * d2)
    d = np.sqrt(d2)
    s5 = math.sqrt(5.0)
    return (1.0 + s5 * d + (5.0 / 3.0) * d2) * np.exp(-s5 * d)


def sqdist(A, B):
    """direct sum_d (a_d - b_d)^2, A: D x n, B: D x m -> n x m"""
    d2 = np.zeros((A.shape[1], B.shape[1]))
    for d in range(A.shape[0]):
        t = A[d][:, None] - B[d][None, :]
        t *= t
        d2 += t
    return d2


class Model:
    """blocks: list of (process, D x n points already divided by the lengthscale);
    cov(process p, process q) = sum of kappa_k over the atoms both contain."""
    ATOMS = {"se": ("se",), "m52": ("m52",), "f1": ("se",), "f2": ("m52",), "f3": ("se", "m52")}

    def __init__(self, blocks):
        self.blocks = blocks
        self.off = np.concatenate([[0], np.cumsum([b[1].shape[1] for b in blocks])]).astype(int)
        self.N = int(self.off[-1])

    def cov_rows_cols(self, r0, r1, c0, c1):
        """dense K[r0:r1, c0:c1] over the concatenated blocks"""
        out = np.zeros((r1 - r0, c1 - c0))
        for I, (p, XI) in enumerate(self.blocks):
            a0, a1 = max(r0, self.off[I]), min(r1, self.off[I + 1])
            if a0 >= a1:
                continue
            for J, (q, XJ) in enumerate(self.blocks):
                b0, b1 = max(c0, self.off[J]), min(c1, self.off[J + 1])
                if b0 >= b1:
                    continue
                common = [k for k in self.ATOMS[p] if k in self.ATOMS[q]]
                if not common:
                    continue
                d2 = sqdist(XI[:, a0 - self.off[I]:a1 - self.off[I]], XJ[:, b0 - self.off[J]:b1 - self.off[J]])
                acc = None
                for k in common:
                    v = kappa("se" if k == "se" else "m52", d2)
                    acc = v if acc is None else acc + v
                out[a0 - r0:a1 - r0, b0 - c0:b1 - c0] = acc
        return out

    def cross(self, proc, Xs):
        """K(x, x*) for x* of process `proc`: N x ns"""
        out = np.zeros((self.N, Xs.shape[1]))
        for I, (p, XI) in enumerate(self.blocks):
            common = [k for k in self.ATOMS[p] if k in self.ATOMS[proc]]
            if not common:
                continue
            d2 = sqdist(XI, Xs)
            out[self.off[I]:self.off[I + 1]] = sum(kappa("se" if k == "se" else "m52", d2) for k in common)
        return out

    def prior_var(self, proc, ns):
        return np.full(ns, float(len(self.ATOMS[proc])))   # kappa(0) = 1 per atom


class WarpModel:
    """h = g1 + 2 g2 - 0.5 g3 of the `w4k` case (module docstring), covariances written out by bilinearity:
      cov(h(x), h(x'))  = ka(u, u') + 4 kb(p, p') + 0.25 ka(us, us') - 0.5 (ka(u, us') + ka(us, u'))
      cov(h(x), g1(x*)) = ka(u, u*) - 0.5 ka(us, u*)
    u = x[0:2] / l, us = (x - s)[0:2] / l, p = [cos(2 pi f x[2]), sin(2 pi f x[2])]; ka = SE, kb = Matern-5/2."""
    L, FREQ, SHIFT = math.sqrt(2.0), 0.3, np.array([0.4, -0.2, 0.1])

    def __init__(self, X):
        self.N = X.shape[1]
        self.u, self.us, self.p = self.views(X)

    def views(self, X):
        u = X[0:2] / self.L
        us = (X - self.SHIFT[:, None])[0:2] / self.L
        w = 2.0 * math.pi * self.FREQ
        p = np.vstack([np.cos(w * X[2]), np.sin(w * X[2])])
        return u, us, p

    def cov_rows_cols(self, r0, r1, c0, c1):
        u, us, p = self.u, self.us, self.p
        ka = lambda A, B: kappa("se", sqdist(A, B))
        return (ka(u[:, r0:r1], u[:, c0:c1]) + 4.0 * kappa("m52", sqdist(p[:, r0:r1], p[:, c0:c1]))
                + 0.25 * ka(us[:, r0:r1], us[:, c0:c1])
                - 0.5 * (ka(u[:, r0:r1], us[:, c0:c1]) + ka(us[:, r0:r1], u[:, c0:c1])))

    def cross(self, proc, Xs):
        assert proc == "g1"
        ustar = Xs[0:2] / self.L
        return kappa("se", sqdist(self.u, ustar)) - 0.5 * kappa("se", sqdist(self.us, ustar))

    def prior_var(self, proc, ns):
        return np.ones(ns)


def assemble_lower(model, s2, pool):
    N = model.N
    A = np.empty((N, N))                     # row-major; only tiles on / below the diagonal are touched
    tiles = [(i, j) for i in range(0, N, TB) for j in range(0, i + 1, TB)]

    def work(ij):
        i, j = ij
        i1, j1 = min(N, i + TB), min(N, j + TB)
        blk = model.cov_rows_cols(i, i1, j, j1)
        if i == j:
            blk[np.diag_indices(i1 - i)] += s2
        A[i:i1, j:j1] = blk

    list(pool.map(work, tiles))
    return A


def cholesky_lower_blocked(A):
    """in-place blocked left-looking Cholesky of the lower triangle of A (row-major N x N)."""
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


def forward_solve(L, B):
    """L^-1 B, blocked, B: N x s (copied)"""
    N = L.shape[0]
    Z = np.array(B, dtype=float, copy=True)
    for k in range(0, N, NB):
        k1 = min(N, k + NB)
        if k > 0:
            Z[k:k1] -= L[k:k1, :k] @ Z[:k]
        Z[k:k1] = sla.solve_triangular(L[k:k1, k:k1], Z[k:k1], lower=True, check_finite=False)
    return Z


def backward_solve(L, B):
    """L^-T B, blocked"""
    N = L.shape[0]
    Z = np.array(B, dtype=float, copy=True)
    starts = list(range(0, N, NB))
    for k in reversed(starts):
        k1 = min(N, k + NB)
        if k1 < N:
            Z[k:k1] -= L[k1:, k:k1].T @ Z[k1:]
        Z[k:k1] = sla.solve_triangular(L[k:k1, k:k1], Z[k:k1], lower=True, trans="T", check_finite=False)
    return Z


def dense_case(name, model, y, proc, Xs, pool):
    t0 = time.time()
    N = model.N
    A = assemble_lower(model, SIGMA2, pool)
    t1 = time.time()
    L = cholesky_lower_blocked(A)
    t2 = time.time()
    Kxs = model.cross(proc, Xs)                              # N x ns
    ZV = forward_solve(L, np.column_stack([y, Kxs]))         # [L^-1 y | V]
    z, V = ZV[:, 0], ZV[:, 1:]
    logdet = 2.0 * np.log(np.diagonal(L)).sum()
    quad = float(z @ z)
    logpdf = -0.5 * (N * LOG2PI + logdet + quad)
    alpha = backward_solve(L, z[:, None])[:, 0]
    mean = Kxs.T @ alpha
    var = model.prior_var(proc, Xs.shape[1]) - np.einsum("ij,ij->j", V, V)
    t3 = time.time()
    print(f"[{name}] N={N} logpdf={logpdf!r}  assemble {t1 - t0:.1f}s chol {t2 - t1:.1f}s "
          f"({N ** 3 / 3 / max(t2 - t1, 1e-9) / 1e9:.0f} GFLOP/s) rest {t3 - t2:.1f}s", flush=True)
    return {"N": N, "logpdf": float(logpdf), "logdet": float(logdet), "quad": quad,
            "post_mean": mean.tolist(), "post_var": var.tolist(),
            "alpha_head": alpha[:8].tolist(), "alpha_norm2": float(alpha @ alpha),
            "cpu_seconds": {"assemble": t1 - t0, "cholesky": t2 - t1, "rest": t3 - t2},
            "cpu_cholesky_gflops": N ** 3 / 3 / max(t2 - t1, 1e-9) / 1e9}


def elbo_case(name, X, y, M, z_noise, pool):
    """App. A.6 with Sigma_y = s2 I, zero mean.  Streams K(z, x) in column chunks."""
    t0 = time.time()
    D, N = X.shape
    Xl = X / math.sqrt(D)
    idx = np.random.default_rng(7).permutation(N)[:M]
    Z = Xl[:, idx]
    Kzz = kappa("se", sqdist(Z, Z))
    Kzz[np.diag_indices(M)] += z_noise
    Lz = sla.cholesky(Kzz, lower=True, check_finite=False)
    sig = math.sqrt(SIGMA2)
    delta = y / sig
    CH = 16384
    chunks = [(c, min(N, c + CH)) for c in range(0, N, CH)]

    def work(c):
        c0, c1 = c
        Kzx = kappa("se", sqdist(Z, Xl[:, c0:c1])) / sig      # M x chunk
        Ac = sla.solve_triangular(Lz, Kzx, lower=True, check_finite=False)
        return Ac @ Ac.T, Ac @ delta[c0:c1], float(np.einsum("ij,ij->", Ac, Ac))

    B = np.eye(M)
    Ad = np.zeros(M)
    frob = 0.0
    for G, v, f in map(work, chunks):      # sequential: BLAS is already threaded, fixed summation order
        B += G
        Ad += v
        frob += f
    Le = sla.cholesky(B, lower=True, check_finite=False)
    w = sla.solve_triangular(Le, Ad, lower=True, check_finite=False)
    logdet_e = 2.0 * np.log(np.diagonal(Le)).sum()
    tmp = N * math.log(SIGMA2) + logdet_e + float(delta @ delta) - float(w @ w)
    dtc = -0.5 * (N * LOG2PI + tmp)
    var_x_sum = float(N) * 1.0                                  # SE prior variance is 1
    elbo = dtc - 0.5 * (var_x_sum / SIGMA2 - frob)
    print(f"[{name}] N={N} M={M} elbo={elbo!r}  {time.time() - t0:.1f}s", flush=True)
    return {"N": N, "M": M, "elbo": float(elbo), "dtc": float(dtc), "frob": frob, "logdet_e": float(logdet_e),
            "cpu_seconds": {"total": time.time() - t0}}


def single(kind, N, D):
    X, y = inputs(N, D)
    return Model([(kind, X / math.sqrt(D))]), y, kind


def gppp3(N, D, lens):
    X, y = inputs(N, D)
    Xl = X / math.sqrt(D)
    cuts = np.concatenate([[0], np.cumsum(lens)]).astype(int)
    assert cuts[-1] == N
    return Model([(p, Xl[:, cuts[i]:cuts[i + 1]]) for i, p in enumerate(("f1", "f2", "f3"))]), y, "f3"


def xs_points(D):
    return np.random.default_rng(987).standard_normal((D, NS)) / math.sqrt(D)


def warp4k():
    X, y = inputs(4096, 3)
    return WarpModel(X), y, "g1"


CASES = {
    "c1": lambda: ("dense", single("se", 2048, 2), 2),
    "n4k": lambda: ("dense", single("m52", 4096, 8), 8),
    "c2": lambda: ("dense", single("se", 16384, 8), 8),
    "c3": lambda: ("dense", gppp3(32768, 4, [10923, 10923, 10922]), 4),
    "c4": lambda: ("elbo", None, 8),
    "c5": lambda: ("dense", single("m52", 65536, 8), 8),
    "n32k": lambda: ("dense", single("m52", 32768, 8), 8),
    "target": lambda: ("dense", gppp3(65536, 8, [21846, 21845, 21845]), 8),
    "w4k": lambda: ("warp", warp4k(), 3),
}


def main():
    names = sys.argv[1:] or list(CASES)
    res = json.load(open(OUT)) if os.path.exists(OUT) else {"generator": "tests/golden/make_baseline_golden.py",
                                                             "sigma2": SIGMA2, "seed": 123456, "xs_seed": 987,
                                                             "cases": {}}
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 1) as pool:
        for name in names:
            kind, built, D = CASES[name]()
            if kind == "elbo":
                X, y = inputs(262144, 8)
                res["cases"][name] = elbo_case(name, X, y, 4096, 1e-6, pool)
            else:
                model, y, proc = built
                # (w4k: the prediction points are the raw standard normals -- the model applies its own transformations)
                Xs = np.random.default_rng(987).standard_normal((D, NS)) if kind == "warp" else xs_points(D)
                res["cases"][name] = dense_case(name, model, y, proc, Xs, pool)
            res["numpy"] = np.__version__
            res["scipy"] = __import__("scipy").__version__
            json.dump(res, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
