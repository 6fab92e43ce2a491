"""Mid-N schedules (round-5 verdict item 5: c2 = 16384 at 0.68, own bar 25.0 ms; gppp3 at 16384: 16.0 ms): sgp_logpdf of the
dense Matern-5/2 GP and the three-block sum model at 12288 / 16384 / 20480 under every schedule the library has for the range
-- whole-matrix dataflow kernel (one / two workgroups per CU), hybrid with panels of 1024 / 2048 / 4096 columns, fewer panel
workgroups (a hybrid panel in pieces was measured through a switch that is gone: profiles/r06_experiments/midn_sweep.md) -- each
on a context of its own (the variables are read at
creation), best of 5 after a warm-up, bit-equality across the variants asserted.  Host-buffer entry point."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
P = entry.load_package()
from stheno_jl_amd import finite_gp as fg  # noqa: E402
from stheno_jl_amd import lib as L  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gpu_hybrid_crossover import problem  # noqa: E402

VARIANTS = [
    ("default", {}),
    ("dataflow-lean", {"SGP_HYBRID": "0", "SGP_DF_FAT_MAX_N": "0"}),
    ("dataflow-fat", {"SGP_HYBRID": "0", "SGP_DF_FAT_MAX_N": "65536"}),
    ("hybrid W2048", {"SGP_HYBRID": "1"}),
    ("hybrid W1024", {"SGP_HYBRID": "1", "SGP_HYBRID_W": "1024"}),
    ("hybrid W4096", {"SGP_HYBRID": "1", "SGP_HYBRID_W": "4096"}),
    ("hybrid W2048 wgs128", {"SGP_HYBRID": "1", "SGP_HYBRID_WGS": "128"}),
    ("hybrid W2048 lean 512", {"SGP_HYBRID": "1", "SGP_HYBRID_FAT": "0", "SGP_HYBRID_WGS": "512"}),
    ("launches look-ahead", {"SGP_HYBRID": "0", "SGP_DATAFLOW": "0"}),
]
KEYS = sorted({k for _, e in VARIANTS for k in e})


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [12288, 16384, 20480]
    res = {}
    for kind in ("dense", "gppp3"):
        for N in sizes:
            fx, y = problem(kind, N)
            spec, m, nk, nbuf = fg._spec_mean_noise(fx)
            m = np.ascontiguousarray(m, dtype=np.float64)
            Y = np.asfortranarray(y.reshape(N, 1))
            rows, vals = {}, []
            for name, env in VARIANTS:
                for k in KEYS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                ctx = L.Context(0)
                out = np.zeros(1)
                ts = []
                for rep in range(6):
                    t0 = time.perf_counter()
                    L.check(ctx.lib.sgp_logpdf(ctx.handle, spec.ref(), L.dptr(m), nk, L.dptr(nbuf), L.dptr(Y), N, 1, L.dptr(out)))
                    ts.append((time.perf_counter() - t0) * 1e3)
                ex, de = ctx.factor_work()
                rows[name] = dict(ms=min(ts[1:]), schedule=ctx.factor_schedule(N), executed=ex / de if de else 1.0)
                vals.append(out[0])
                ctx.close()
            assert all(v == vals[0] for v in vals), (kind, N, vals)
            best = min(rows, key=lambda k: rows[k]["ms"])
            res[f"{kind}_{N}"] = dict(rows=rows, best=best, bit_equal=True)
            print(f"{kind} N={N}: " + " | ".join(f"{k} {v['ms']:.2f}" for k, v in rows.items()) + f" || best: {best}", flush=True)
    for k in KEYS:
        os.environ.pop(k, None)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
