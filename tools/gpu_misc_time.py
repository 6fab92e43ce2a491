"""Timing sweep of the remaining host-API rows (rand, sparse posterior, kernelmatrix, marginals) --
run under rocprofv3 --kernel-trace --stats to spot kernels that are out of proportion."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
P = entry.load_package()
rng = np.random.default_rng(0)
D = 8
f = 1.3 * P.atomic(P.GP(P.Matern52Kernel()), P.GPC())


def timed(label, fn, reps=2):
    for _ in range(reps):
        t0 = time.time(); out = fn(); t1 = time.time()
    print(f"{label}: {1e3 * (t1 - t0):.1f} ms", flush=True)
    return out


N = 16384
X = P.ColVecs(rng.standard_normal((D, N)) / np.sqrt(D))
y = rng.standard_normal(N)
timed(f"rand N={N} S=256", lambda: P.rand(np.random.default_rng(1), f(X, 0.1), 256))
timed(f"marginals N={N}", lambda: P.marginals(f(X, 0.1)))
Xk = P.ColVecs(rng.standard_normal((D, 8192)) / np.sqrt(D))
timed("kernelmatrix 8192^2 (host out)", lambda: P.prior_cov(f, Xk))
N, M, Ns = 65536, 2048, 16384
X = P.ColVecs(rng.standard_normal((D, N)) / np.sqrt(D))
Z = P.ColVecs(rng.standard_normal((D, M)) / np.sqrt(D))
Xs = P.ColVecs(rng.standard_normal((D, Ns)) / np.sqrt(D))
y = rng.standard_normal(N)
post = timed(f"posterior(VFE) N={N} M={M}", lambda: P.posterior(P.VFE(f(Z, 1e-6)), f(X, 0.1), y))
timed(f"sparse mean_and_var Ns={Ns}", lambda: P.mean_and_var(post(Xs, 0.0)))
# input dimension beyond the templated kernels (assemble_bigd_kernel): logpdf at N = 16384 for D = 8 / 64 / 256 --
# the difference to D = 8 is what the assembly costs
for Dh in (8, 64, 256):
    Xh = P.ColVecs(rng.standard_normal((Dh, 16384)) / np.sqrt(Dh))
    yh = rng.standard_normal(16384)
    timed(f"logpdf N=16384 D={Dh}", lambda: P.logpdf(f(Xh, 0.1), yh))
