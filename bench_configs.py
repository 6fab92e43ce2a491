"""The BASELINE.json workloads, built once for bench.py and the GPU golden tests.

Every configuration uses rng = default_rng(123456); X = rng.standard_normal((D, N)) stored as a
ColVecs matrix (D x N column-major, the layout of /root/reference/test/gp/util.jl:24);
y = rng.standard_normal(N); zero mean; lengthscale sqrt(D); sigma^2 = 0.1 (test/gp/util.jl:82).
tests/golden/make_baseline_golden.py states the same inputs independently (NumPy only) and holds
the CPU values of every configuration in tests/golden/baseline_configs.json.
"""
from __future__ import annotations

import json
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(ROOT, "tests", "golden", "baseline_configs.json")
SIGMA2 = 0.1

CONFIGS = {
    # name: (kind, N, D)                  BASELINE.json configs[0..4] = c1..c5
    "c1": ("se", 2048, 2),
    "c2": ("se", 16384, 8),
    "c3": ("gppp3", 32768, 4),      # @gppp f3 = f1 + f2 over BlockData (:f1,10923),(:f2,10923),(:f3,10922)
    "c4": ("elbo", 262144, 8),      # sparse ELBO, M = 4096 inducing points
    "c5": ("matern52", 65536, 8),
    "target": ("gppp3", 65536, 8),  # north-star run: c3's model, blocks 21846 / 21845 / 21845
    "n4k": ("matern52", 4096, 8),
    "n32k": ("matern52", 32768, 8),
    # every input transformation of compose.jl in one programme (tests/golden/make_baseline_golden.py `w4k`)
    "w4k": ("warp", 4096, 3),
}
GPPP_BLOCKS = {"c3": [10923, 10923, 10922], "target": [21846, 21845, 21845]}
ELBO_M = 4096
ELBO_ZNOISE = 1e-6


def make_inputs(N, D):
    rng = np.random.default_rng(123456)
    X = np.asfortranarray(rng.standard_normal((D, N)))
    y = rng.standard_normal(N)
    return X, y


def xs_points(D, ns=64):
    """the 64 prediction points of the goldens (unscaled; the lengthscale is applied like X's)"""
    return np.asfortranarray(np.random.default_rng(987).standard_normal((D, ns)))


def describe(name):
    kind, N, D = CONFIGS[name]
    body = {"warp": f"@gppp h = g1 + 2 g2 - 0.5 g3 with select / stretch / periodic / shift views of two atoms, N={N}, D={D}",
            "gppp3": f"@gppp f3=f1+f2 (SE + Matern52) over 3 BlockData blocks {GPPP_BLOCKS.get(name)}, total N={N}, D={D}",
            "elbo": f"sparse ELBO, SE, M={ELBO_M} inducing points, N={N}, D={D}"}.get(
                kind, f"single GP, {kind}, N={N}, D={D}")
    return body + f", lengthscale sqrt(D), sigma2={SIGMA2} (BASELINE config '{name}')"


def build(pkg, name):
    """-> dict(kind, N, D, y, fx [, f, x, vfe, xs_new]) built with the product's host mirror `pkg`."""
    kind, N, D = CONFIGS[name]
    X, y = make_inputs(N, D)
    out = dict(kind=kind, N=N, D=D, y=y, X=X)
    ls = math.sqrt(D)
    if kind == "gppp3":
        F = pkg.gppp_sum_model()
        cuts = np.concatenate([[0], np.cumsum(GPPP_BLOCKS[name])]).astype(int)
        xb = pkg.BlockData([pkg.GPPPInput(k, pkg.ColVecs(np.asfortranarray(X[:, cuts[i]:cuts[i + 1]] / ls)))
                            for i, k in enumerate(("f1", "f2", "f3"))])
        out.update(f=F, x=xb, fx=F(xb, SIGMA2),
                   xs_new=pkg.GPPPInput("f3", pkg.ColVecs(np.asfortranarray(xs_points(D) / ls))))
    elif kind == "warp":
        def prog(GP):
            a, b = GP(pkg.SEKernel()), GP(pkg.Matern52Kernel())
            g1 = pkg.select(pkg.stretch(a, 1.0 / math.sqrt(2.0)), [0, 1])
            g2 = pkg.select(pkg.periodic(b, 0.3), 2)
            g3 = pkg.shift(g1, np.array([0.4, -0.2, 0.1]))
            return {"g1": g1, "g2": g2, "g3": g3, "h": g1 + 2.0 * g2 - 0.5 * g3}
        F = pkg.gppp(prog)
        xh = pkg.GPPPInput("h", pkg.ColVecs(X))
        out.update(f=F, x=xh, fx=F(xh, SIGMA2),
                   xs_new=pkg.GPPPInput("g1", pkg.ColVecs(np.asfortranarray(np.random.default_rng(987).standard_normal((D, 64))))))
    elif kind == "elbo":
        f = pkg.stretch(pkg.atomic(pkg.GP(pkg.SEKernel()), pkg.GPC()), 1.0 / ls)
        Z = np.asfortranarray(X[:, np.random.default_rng(7).permutation(N)[:ELBO_M]])
        fx, fz = f(pkg.ColVecs(X), SIGMA2), f(pkg.ColVecs(Z), ELBO_ZNOISE)
        out.update(f=f, x=pkg.ColVecs(X), fx=fx, fz=fz, vfe=pkg.VFE(fz))
    else:
        k = {"se": pkg.SEKernel, "matern52": pkg.Matern52Kernel}[kind]()
        f = pkg.stretch(pkg.atomic(pkg.GP(k), pkg.GPC()), 1.0 / ls)
        out.update(f=f, x=pkg.ColVecs(X), fx=f(pkg.ColVecs(X), SIGMA2), xs_new=pkg.ColVecs(xs_points(D)))
    return out


_golden_cache = None


def golden(name):
    """CPU known-answer record of config `name` (None when the JSON does not hold it)."""
    global _golden_cache
    if _golden_cache is None:
        _golden_cache = json.load(open(GOLDEN)) if os.path.exists(GOLDEN) else {"cases": {}}
    return _golden_cache["cases"].get(name)
