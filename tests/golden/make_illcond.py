"""Generate tests/golden/illcond_truth.json: 60-digit (mpmath) logpdf values of a zero-mean SE-kernel
GP on sorted 1-D inputs with tiny observation noise -- covariances with condition numbers 1e6..1e12,
where an unstable triangular solve shows up as lost digits or a spurious PosDefException.
Inputs are stored with the answers so the fixture does not depend on an RNG implementation.

    python tests/golden/make_illcond.py      (about two minutes)
"""
import json
import os

import mpmath as mp
import numpy as np

mp.mp.dps = 60
CASES = [(200, 1e-6), (200, 1e-9), (200, 1e-12), (400, 1e-8), (130, 1e-10)]


def truth(x, y, s2):
    n = len(x)
    K = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            K[i, j] = mp.exp(-(mp.mpf(float(x[i])) - mp.mpf(float(x[j]))) ** 2 / 2)
        K[i, i] += mp.mpf(s2)
    L = mp.cholesky(K)
    z = mp.lu_solve(L, mp.matrix([float(v) for v in y]))
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(n))
    return -(n * mp.log(2 * mp.pi) + logdet + sum(v * v for v in z)) / 2


def main():
    rng = np.random.default_rng(1)
    out = []
    for n, s2 in CASES:
        x = np.sort(rng.uniform(-3, 3, n))
        y = rng.standard_normal(n)
        t = truth(x, y, s2)
        out.append({"N": n, "noise": s2, "x": x.tolist(), "y": y.tolist(), "logpdf": mp.nstr(t, 25)})
        print(n, s2, mp.nstr(t, 20), flush=True)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "illcond_truth.json")
    with open(path, "w") as f:
        json.dump({"kernel": "SEKernel", "mean": 0.0, "cases": out}, f)


if __name__ == "__main__":
    main()
