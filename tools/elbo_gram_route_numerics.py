"""Round-4 verdict, item 2: "halve the ELBO's flops where K(z,z) allows it" -- form G = Kzx Lambda^-1 Kxz with ONE M^2 N
product and get B = I + Lz^-1 G Lz^-T with two M^3 triangular solves, instead of A = Lz^-1 Kzx Lambda^-1/2 (M^2 N) and A A'
(M^2 N).  The Gram route amplifies rounding by cond(K(z,z)) instead of its square root; the verdict asked for a gate
(eps * cond * safety <= 1e-10).  This script measures, in NumPy fp64, what the two routes give on the benchmark's own model
family (SE kernel, lengthscale sqrt(D), inducing points = a random subset of the data, Sigma_z = 1e-6 I, sigma^2 = 0.1) and on
1-D / 2-D cases: the relative difference of the two ELBO values, cond(K(z,z)) and the cheap estimate
trace(K(z,z)) / min pivot^2 a gate could use.  CPU only; NumPy / SciPy.  Result: profiles/r05_experiments/elbo_gram_route.md."""
import math

import numpy as np
import scipy.linalg as sla


def elbo_routes(X, Z, y, s2, zn, kern):
    N, M = X.shape[1], Z.shape[1]

    def K(A, B):
        d2 = (A * A).sum(0)[:, None] + (B * B).sum(0)[None, :] - 2 * A.T @ B
        return kern(np.maximum(d2, 0))
    Kzz = K(Z, Z) + zn * np.eye(M)
    Kzx = K(Z, X)
    Lz = sla.cholesky(Kzz, lower=True)
    w = 1.0 / s2
    d = y * math.sqrt(w)

    def fin(Lb, c, trAA):
        return (-0.5 * (N * math.log(2 * math.pi) + N * math.log(s2) + 2 * np.log(np.diag(Lb)).sum() + d @ d - c @ c)
                - 0.5 * (N * 1.0 / s2 - trAA))
    # route A (the library's): A = Lz^-1 Kzx Lambda^-1/2, B = I + A A'
    A = sla.solve_triangular(Lz, Kzx, lower=True) * math.sqrt(w)
    Lb = sla.cholesky(np.eye(M) + A @ A.T, lower=True)
    eA = fin(Lb, sla.solve_triangular(Lb, A @ d, lower=True), (A * A).sum())
    # route B (Gram first): G = Kzx Lambda^-1 Kxz, S = Lz^-1 G Lz^-T
    G = (Kzx * w) @ Kzx.T
    T = sla.solve_triangular(Lz, G, lower=True)
    S = sla.solve_triangular(Lz, T.T, lower=True)
    S = 0.5 * (S + S.T)
    Lb2 = sla.cholesky(np.eye(M) + S, lower=True)
    u = sla.solve_triangular(Lz, Kzx @ (d * math.sqrt(w)), lower=True)
    eB = fin(Lb2, sla.solve_triangular(Lb2, u, lower=True), np.trace(S))
    est = np.trace(Kzz) / (np.diag(Lz) ** 2).min()
    return eA, eB, np.linalg.cond(Kzz), est


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    se = lambda d2: np.exp(-0.5 * d2)
    print("| D | N | M | lengthscale | Sigma_z | cond K(z,z) | trace / min pivot^2 | eps x estimate x 8 | |ELBO_A - ELBO_B| / |ELBO| |")
    print("|---|---|---|---|---|---|---|---|---|")
    for (D, N, M, ls, zn) in [(8, 16384, 512, math.sqrt(8), 1e-6), (8, 16384, 1024, math.sqrt(8), 1e-6),
                              (8, 16384, 2048, math.sqrt(8), 1e-6), (8, 16384, 1024, math.sqrt(8), 1e-2),
                              (1, 4096, 256, 1.0, 1e-6), (2, 8192, 512, 1.0, 1e-6), (1, 4096, 64, 0.3, 1e-6)]:
        X = rng.standard_normal((D, N)) / ls
        Z = X[:, rng.permutation(N)[:M]]
        y = rng.standard_normal(N)
        eA, eB, cond, est = elbo_routes(X, Z, y, 0.1, zn, se)
        print(f"| {D} | {N} | {M} | {ls:.3g} | {zn:g} | {cond:.2e} | {est:.2e} | {2.2e-16 * est * 8:.1e} | {abs(eA - eB) / abs(eA):.1e} |")
