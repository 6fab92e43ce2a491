"""The data-sharded ELBO gradient of the multi-GPU context at a realistic size (N = 131 072, M = 2048) on loopback ranks against the
single-GPU result: time per call and the largest relative difference over every result.  Usage: python tools/gpu_elbo_grad_sharded_check.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
P = entry.load_package()
rng = np.random.default_rng(3)
D, N, M = 8, 131072, 2048
f = P.stretch(P.atomic(P.GP(P.SEKernel()), P.GPC()), 1.0 / np.sqrt(D))
X = np.asfortranarray(rng.standard_normal((D, N)))
Z = np.asfortranarray(X[:, rng.permutation(N)[:M]] + 0.01)
y = rng.standard_normal(N)
vfe, fx = P.VFE(f(P.ColVecs(Z), 1e-4)), f(P.ColVecs(X), 0.1)
def run(ctx, kw):
    prev = P.lib.set_default_context(ctx)
    try:
        g = P.elbo_and_gradient(vfe, fx, y, **kw)
        t0 = time.perf_counter(); g = P.elbo_and_gradient(vfe, fx, y, **kw); dt = time.perf_counter() - t0
    finally:
        P.lib.set_default_context(prev)
    return g, dt
c1 = P.lib.Context(0)
for kw in (dict(), dict(inputs=True)):
    g0, t0 = run(c1, kw)
    for nr in (2, 8):
        cm = P.lib.Context(devices=[0] * nr)
        g1, t1 = run(cm, kw)
        cm.close()
        err = max(abs(g1["elbo"] - g0["elbo"]) / abs(g0["elbo"]),
                  np.abs(g1["y"] - g0["y"]).max() / np.abs(g0["y"]).max(),
                  max(np.abs(a - b).max() / max(1.0, np.abs(a).max()) for a, b in zip(g0["_raw"]["xz"], g1["_raw"]["xz"])),
                  max(np.abs(a - b).max() / max(1.0, np.abs(a).max()) for a, b in zip(g0["_raw"]["zz"], g1["_raw"]["zz"])))
        if kw:
            err = max(err, max(np.abs(a - b).max() / max(1.0, np.abs(a).max()) for a, b in zip(g0["x"], g1["x"])),
                      max(np.abs(a - b).max() / max(1.0, np.abs(a).max()) for a, b in zip(g0["z"], g1["z"])))
        print(f"N={N} M={M} {kw}: single {t0*1e3:.1f} ms, {nr} loopback ranks {t1*1e3:.1f} ms, max rel diff {err:.2e}", flush=True)
