import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
P = entry.load_package()
for N in [int(a) for a in sys.argv[1:]] or [8192]:
    rng = np.random.default_rng(0)
    D = 8
    X = P.ColVecs(rng.standard_normal((D, N)) / np.sqrt(D))
    y = rng.standard_normal(N)
    f = 1.3 * P.atomic(P.GP(P.Matern52Kernel()), P.GPC())
    fx = f(X, 0.1)
    for it in range(2):
        t0 = time.time(); lp = P.logpdf(fx, y); t1 = time.time()
        g = P.logpdf_and_gradient(fx, y); t2 = time.time()
    print(f"N={N}: logpdf {1e3*(t1-t0):.1f} ms, logpdf+grad {1e3*(t2-t1):.1f} ms (host API incl. alloc/copies), "
          f"d_coef={g['terms'][0]['d_coef']:.6f} d_noise={g['noise']:.6f}", flush=True)

# ELBO and its gradient (VFE, M inducing points)
for N, M in [(65536, 2048)]:
    rng = np.random.default_rng(1)
    D = 8
    X = P.ColVecs(rng.standard_normal((D, N)) / np.sqrt(D))
    Z = P.ColVecs(rng.standard_normal((D, M)) / np.sqrt(D))
    y = rng.standard_normal(N)
    f = 1.3 * P.atomic(P.GP(P.Matern52Kernel()), P.GPC())
    for it in range(2):
        t0 = time.time(); e = P.elbo(P.VFE(f(Z, 1e-6)), f(X, 0.1), y); t1 = time.time()
        g = P.elbo_and_gradient(P.VFE(f(Z, 1e-6)), f(X, 0.1), y); t2 = time.time()
    print(f"N={N} M={M}: elbo {1e3*(t1-t0):.1f} ms, elbo+grad {1e3*(t2-t1):.1f} ms (host API incl. alloc/copies), "
          f"elbo={e:.6f}/{g['elbo']:.6f}", flush=True)
