"""Operators of the multi-GPU context (round 3: posterior on the kept sharded factor, rand, data-sharded ELBO) with
P loopback ranks on the ONE GPU of the box, next to the single-GPU driver: what the sharded code paths cost when they
cannot gain anything (same GPU), and their agreement.  usage: gpu_multi_ops_time.py [N=16384] [P=4]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

P = g.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(1)
D = 8
f = P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 1.0 / np.sqrt(D))
X = np.asfortranarray(rng.standard_normal((D, N)))
y = rng.standard_normal(N)
xs = P.ColVecs(np.asfortranarray(rng.standard_normal((D, 1024))))
Z = np.asfortranarray(rng.standard_normal((N, 8)))
fx = f(P.ColVecs(X), 0.1)
M = 2048
fz = f(P.ColVecs(np.asfortranarray(X[:, :M] + 0.01)), 1e-6)


def timed(fn, n=2):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    return (time.perf_counter() - t0) / n * 1e3, out


res = {}
for tag, devs in (("single", None), (f"{R} loopback ranks", [0] * R)):
    ctx = P.lib.Context(devices=devs) if devs else None
    prev = P.lib.set_default_context(ctx) if ctx else None
    try:
        t_post, post = timed(lambda: P.posterior(fx, y), 1)
        t_pred, mv = timed(lambda: post.mean_and_var(xs))
        t_rand, r = timed(lambda: P.rand(None, fx, 8, Z=Z))
        t_elbo, e = timed(lambda: P.elbo(P.VFE(fz), fx, y))
        res[tag] = (mv, r, e)
        print(f"{tag:18s} N={N}: posterior(fx, y) {t_post:8.1f} ms | mean_and_var at 1024 points {t_pred:7.1f} ms | "
              f"rand S=8 {t_rand:7.1f} ms | elbo M={M} {t_elbo:7.1f} ms", flush=True)
        del post
    finally:
        if ctx:
            P.lib.set_default_context(prev)
            ctx.close()
(mv0, r0, e0), (mv1, r1, e1) = res.values()
print("agreement sharded vs single: mean %.1e var %.1e rand %.1e elbo %.1e" % (
    np.max(np.abs(mv1[0] - mv0[0])) / np.max(np.abs(mv0[0])), np.max(np.abs(mv1[1] - mv0[1])),
    np.max(np.abs(r1 - r0)) / np.max(np.abs(r0)), abs(e1 - e0) / abs(e0)))
