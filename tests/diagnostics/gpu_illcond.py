"""How does the explicit-inverse panel TRSM behave on ill-conditioned covariances?  Compare the HIP
logpdf and the oracle (LAPACK) with a 50-digit mpmath reference."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
P = entry.load_package()
import oracle.abstractgps as agp, oracle.kernelfunctions as kf, oracle.stheno as st
import mpmath as mp
mp.mp.dps = 60
rng = np.random.default_rng(1)
for N, s2 in [(200, 1e-6), (200, 1e-9), (200, 1e-12), (400, 1e-8), (130, 1e-10)]:
    x = np.sort(rng.uniform(-3, 3, N))
    y = rng.standard_normal(N)
    fo = st.atomic(agp.GP(kf.SEKernel()), st.GPC())
    fp = P.atomic(P.GP(P.SEKernel()), P.GPC())
    try:
        lo = agp.logpdf(fo(x, s2), y)
    except Exception as e:
        lo = float("nan")
    try:
        lp = P.logpdf(fp(x, s2), y)
    except Exception as e:
        lp = float("nan"); print("gpu exc", e)
    K = mp.matrix(N, N)
    for i in range(N):
        for j in range(N):
            K[i, j] = mp.exp(-(mp.mpf(x[i]) - mp.mpf(x[j])) ** 2 / 2)
        K[i, i] += mp.mpf(s2)
    L = mp.cholesky(K)
    z = mp.lu_solve(L, mp.matrix(list(y)))
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(N))
    truth = -(N * mp.log(2 * mp.pi) + logdet + sum(v * v for v in z)) / 2
    t = float(truth)
    # posterior mean / var at a few points against the oracle (both LAPACK-accurate at best)
    try:
        xs = np.linspace(-2.5, 2.5, 7)
        po = agp.posterior(fo(x, s2), y); pp = P.posterior(fp(x, s2), y)
        mo, vo = po.mean_and_var(xs) if hasattr(po, "mean_and_var") else (po.mean(xs), po.var(xs))
        mq, vq = P.mean_and_var(pp(xs, 0.0)) if hasattr(P, "mean_and_var") else (None, None)
        # 60-digit posterior mean
        Kx = mp.matrix(N, N)
        for i in range(N):
            for j in range(N):
                Kx[i, j] = mp.exp(-(mp.mpf(x[i]) - mp.mpf(x[j])) ** 2 / 2)
            Kx[i, i] += mp.mpf(s2)
        al = mp.lu_solve(Kx, mp.matrix(list(y)))
        mt = np.array([float(sum(mp.exp(-(mp.mpf(v) - mp.mpf(x[i])) ** 2 / 2) * al[i] for i in range(N))) for v in xs])
        print(f"   posterior mean rel err vs truth: oracle {np.max(np.abs(mo-mt)/np.abs(mt)):.2e}  gpu {np.max(np.abs(mq-mt)/np.abs(mt)):.2e}   var max abs diff {np.max(np.abs(vq-vo)):.2e}")
    except Exception as e:
        print("   posterior exc", repr(e)[:200])
    print(f"N={N} s2={s2:g}: truth {t:.10e}  oracle rel err {abs(lo-t)/abs(t):.2e}  gpu rel err {abs(lp-t)/abs(t):.2e}", flush=True)
