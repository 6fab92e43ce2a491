"""Prediction-side path: posterior(fx, y) then marginals at many test points."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
P = entry.load_package()
for N, Ns in [(int(a.split("x")[0]), int(a.split("x")[1])) for a in sys.argv[1:]] or [(16384, 16384)]:
    rng = np.random.default_rng(0)
    D = 8
    X = P.ColVecs(rng.standard_normal((D, N)) / np.sqrt(D))
    Xs = P.ColVecs(rng.standard_normal((D, Ns)) / np.sqrt(D))
    y = rng.standard_normal(N)
    f = 1.3 * P.atomic(P.GP(P.Matern52Kernel()), P.GPC())
    for it in range(2):
        t0 = time.time(); post = P.posterior(f(X, 0.1), y); t1 = time.time()
        m, v = P.mean_and_var(post(Xs, 0.0)); t2 = time.time()
    print(f"N={N} Ns={Ns}: posterior {1e3*(t1-t0):.1f} ms, mean_and_var {1e3*(t2-t1):.1f} ms "
          f"(host API incl. copies)  mean[0]={m[0]:.6f} var[0]={v[0]:.6f}", flush=True)
