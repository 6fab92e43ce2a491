"""Size sweep of the dataflow factorisation against the launch-based schedules: logpdf of one Matern-5/2 GP (D = 8)
through the host-buffer C-ABI (sgp_logpdf), best of `reps` calls per variant.  usage: gpu_df_sizes.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

P = g.load_package()
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 6144, 8192, 12288, 16384, 24576]
VARIANTS = [
    ("launches", {"SGP_DATAFLOW": "0"}),
    ("dataflow 2 WG/CU", {"SGP_DATAFLOW": "1", "SGP_DF_FAT_MAX_N": "0"}),
    ("dataflow 1 WG/CU", {"SGP_DATAFLOW": "1", "SGP_DF_FAT_MAX_N": "1000000"}),
    ("default", {}),
]
KEYS = sorted({k for _, e in VARIANTS for k in e})
print(f"{'N':>7s} " + " ".join(f"{n:>17s}" for n, _ in VARIANTS) + "   (ms, best of 5; logpdf identical across variants: checked)")
for N in sizes:
    rng = np.random.default_rng(N)
    x = P.ColVecs(np.asfortranarray(rng.standard_normal((8, N))))
    f = P.atomic(P.GP(P.with_lengthscale(P.Matern52Kernel(), np.sqrt(8.0))), P.GPC())
    y = rng.standard_normal(N)
    row, vals = [], []
    for name, env in VARIANTS:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["SGP_DF_TIMEOUT_S"] = "3"
        ctx = P.lib.Context(0)
        prev = P.lib.set_default_context(ctx)
        try:
            best = 1e9
            for rep in range(7):
                t0 = time.perf_counter()
                v = P.logpdf(f(x, 0.1), y)
                dt = (time.perf_counter() - t0) * 1e3
                if rep >= 2:
                    best = min(best, dt)
            row.append(best)
            vals.append(v)
        finally:
            P.lib.set_default_context(prev)
            ctx.close()
    assert all(v == vals[0] for v in vals), vals
    print(f"{N:7d} " + " ".join(f"{b:17.3f}" for b in row), flush=True)
