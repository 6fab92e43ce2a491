"""Generates tests/golden/grad_configs.json: CPU known-answer GRADIENTS of logpdf for the BASELINE sizes the gradient is
benchmarked at (bench.py `grad` extra; SURVEY.md 8f-1: the reference's main use, hyper-parameter learning through Zygote,
/root/reference/examples/getting_started/script.jl:154-213).

Standalone like make_baseline_golden.py: NumPy / SciPy only, imports nothing from this repository.  For a zero-mean GP with
C = K + s2 I, alpha = C^-1 y:
    G = d logpdf / d C = (alpha alpha' - C^-1) / 2
    d logpdf / d s2      = tr(G)
    d logpdf / d inscale = sum_ij G_ij dK_ij / dg |_{g = 1},   K_ij = kappa(g |x_i - x_j|)   (x already divided by the lengthscale:
                           `stretch(f, g)` on top of the model, the derivative the library reports per term as d_inscale)
      SE        : dK / dg = -d^2 exp(-d^2 / 2)
      Matern-5/2: dK / dg = -(5/3) d^2 (1 + sqrt5 d) exp(-sqrt5 d)
Configurations (inputs as bench_configs.make_inputs: default_rng(123456), lengthscale sqrt(D), s2 = 0.1):
    n4k  Matern-5/2, N = 4096,  D = 8        c2  SE, N = 16384, D = 8
C^-1 by plain solve against the identity (cho_solve), N <= 16384.

  python tests/golden/make_grad_golden.py            # ~3 min on 8 cores
"""
import json
import math
import os
import sys
import time

import numpy as np
import scipy.linalg as sla

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "grad_configs.json")
SIGMA2 = 0.1
CASES = {"n4k": ("matern52", 4096, 8), "c2": ("se", 16384, 8)}


def main(names):
    rec = json.load(open(OUT)) if os.path.exists(OUT) else {"cases": {}}
    for name in names:
        kind, N, D = CASES[name]
        t0 = time.time()
        rng = np.random.default_rng(123456)
        X = rng.standard_normal((D, N)) / math.sqrt(D)
        y = rng.standard_normal(N)
        d2 = np.zeros((N, N))
        for d in range(D):
            t = X[d][:, None] - X[d][None, :]
            t *= t
            d2 += t
        if kind == "se":
            K = np.exp(-0.5 * d2)
            dK = -d2 * K
        else:
            r = np.sqrt(d2)
            e = np.exp(-math.sqrt(5.0) * r)
            K = (1.0 + math.sqrt(5.0) * r + (5.0 / 3.0) * d2) * e
            dK = -(5.0 / 3.0) * d2 * (1.0 + math.sqrt(5.0) * r) * e
            del r, e
        del d2
        K[np.diag_indices(N)] += SIGMA2
        Lc = sla.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
        del K
        alpha = sla.cho_solve((Lc, True), y, check_finite=False)
        logdet = 2.0 * np.sum(np.log(np.diag(Lc)))
        logpdf = -0.5 * (N * math.log(2.0 * math.pi) + logdet + float(y @ alpha))
        Cinv = sla.cho_solve((Lc, True), np.eye(N), overwrite_b=True, check_finite=False)
        del Lc
        d_s2 = 0.5 * (float(alpha @ alpha) - float(np.trace(Cinv)))
        # sum_ij G_ij dK_ij = (alpha' dK alpha - sum(C^-1 .* dK)) / 2
        d_g = 0.5 * (float(alpha @ (dK @ alpha)) - float(np.sum(Cinv * dK)))
        rec["cases"][name] = {"kind": kind, "N": N, "D": D, "sigma2": SIGMA2, "logpdf": logpdf, "d_sigma2": d_s2,
                              "d_inscale": d_g, "d_y_norm": float(np.linalg.norm(alpha)), "seconds": time.time() - t0}
        print(name, rec["cases"][name], flush=True)
        del Cinv, dK
    rec["generator"] = "tests/golden/make_grad_golden.py (NumPy/SciPy, standalone)"
    json.dump(rec, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))
