"""A/B of dataflow-factorisation knobs: logpdf of one Matern-5/2 GP (D = 8) through the host-buffer C-ABI, best of 3 after
2 warm-up calls, one context per variant.  usage: gpu_df_variants.py N 'NAME:K=V,K=V' ..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

P = g.load_package()
N = int(sys.argv[1])
V = []
for a in sys.argv[2:]:
    name, _, kv = a.partition(":")
    V.append((name, dict(x.split("=") for x in kv.split(",") if x)))
keys = sorted({k for _, e in V for k in e})
rng = np.random.default_rng(N)
x = P.ColVecs(np.asfortranarray(rng.standard_normal((8, N))))
f = P.atomic(P.GP(P.with_lengthscale(P.Matern52Kernel(), np.sqrt(8.0))), P.GPC())
y = rng.standard_normal(N)
vals = []
for name, env in V:
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ.setdefault("SGP_DF_TIMEOUT_S", "5")
    ctx = P.lib.Context(0)
    prev = P.lib.set_default_context(ctx)
    try:
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            v = P.logpdf(f(x, 0.1), y)
            ts.append((time.perf_counter() - t0) * 1e3)
        vals.append(v)
        print(f"N={N:6d} {name:22s} best {min(ts[2:]):9.3f} ms  (all: {' '.join('%.2f' % t for t in ts)})", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"N={N:6d} {name:22s} FAILED: {e}", flush=True)
    finally:
        P.lib.set_default_context(prev)
        ctx.close()
print(f"N={N}: logpdf identical across variants: {all(v == vals[0] for v in vals)} ({vals[0]!r})", flush=True)
