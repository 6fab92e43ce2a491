"""Task order of the dataflow factorisation (chol_df.hip / df_order.h): column-major ids against the XCD-affine queues over
patch shapes, and the launch-based schedule beside them.  logpdf of one Matern-5/2 GP (D = 8) through the host-buffer
C-ABI, best of 3 after 2 warm-up calls.  usage: gpu_df_order.py [N ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

P = g.load_package()
sizes = [int(a) for a in sys.argv[1:]] or [16384, 32768, 65536]
BIG = "1000000000"


def variants(N):
    v = [("launches", {"SGP_DATAFLOW": "0"})]
    for fat in ((1, 0) if N <= 24576 else (0,)):
        base = {"SGP_DATAFLOW": "1", "SGP_DF_FAT_MAX_N": BIG if fat else "0"}
        tag = "fat" if fat else "lean"
        v.append((f"{tag} colmajor", dict(base, SGP_DF_ORDER="0")))
        slots = 32 if fat else 64
        for pc in (1, 2, 4, 8):
            v.append((f"{tag} q{slots // pc}x{pc}", dict(base, SGP_DF_ORDER="1", SGP_DF_PR=str(slots // pc), SGP_DF_PC=str(pc))))
        if not fat:
            v.append((f"{tag} q32x4", dict(base, SGP_DF_ORDER="1", SGP_DF_PR="32", SGP_DF_PC="4")))
            v.append((f"{tag} q8x4", dict(base, SGP_DF_ORDER="1", SGP_DF_PR="8", SGP_DF_PC="4")))
    return v


for N in sizes:
    rng = np.random.default_rng(N)
    x = P.ColVecs(np.asfortranarray(rng.standard_normal((8, N))))
    f = P.atomic(P.GP(P.with_lengthscale(P.Matern52Kernel(), np.sqrt(8.0))), P.GPC())
    y = rng.standard_normal(N)
    vals = []
    V = variants(N)
    keys = sorted({k for _, e in V for k in e})
    for name, env in V:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["SGP_DF_TIMEOUT_S"] = "5"
        ctx = P.lib.Context(0)
        prev = P.lib.set_default_context(ctx)
        try:
            ts = []
            for rep in range(5):
                t0 = time.perf_counter()
                v = P.logpdf(f(x, 0.1), y)
                ts.append((time.perf_counter() - t0) * 1e3)
            vals.append(v)
            print(f"N={N:6d} {name:18s} best {min(ts[2:]):9.3f} ms  (all: {' '.join('%.2f' % t for t in ts)})  logpdf {v!r}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"N={N:6d} {name:18s} FAILED: {e}", flush=True)
        finally:
            P.lib.set_default_context(prev)
            ctx.close()
    print(f"N={N}: logpdf identical across variants: {all(v == vals[0] for v in vals)}", flush=True)
