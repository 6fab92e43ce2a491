"""Where the hybrid schedule (capi.hip: use_hybrid) takes over from the dataflow kernel: sgp_logpdf of a dense Matern-5/2 GP and
of the three-block sum model at sizes between 16384 and 40960, under SGP_HYBRID = 0 / 1 (each context reads the variable when
it is created).  Host-buffer entry point: the D x N inputs travel, the N x N matrix never does.  Run on the GPU box."""
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


def problem(kind, N, D=8):
    rng = np.random.default_rng(N)
    if kind == "dense":
        f = P.atomic(P.GP(P.Matern52Kernel()), P.GPC())
        fx = f(P.ColVecs(np.asfortranarray(rng.standard_normal((D, N)))), 0.1)
    else:
        F = P.gppp_sum_model()
        n1 = N // 3
        xs = [np.asfortranarray(rng.standard_normal((D, n))) for n in (n1, n1, N - 2 * n1)]
        fx = F(P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)]), 0.1)
    return fx, rng.standard_normal(N)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16384, 20480, 24576, 28672, 32768, 40960]
    for kind in ("dense", "gppp3"):
        for N in sizes:
            fx, y = problem(kind, N)
            spec, m, nk, nbuf = fg._spec_mean_noise(fx)
            m = np.ascontiguousarray(m, dtype=np.float64)
            Y = np.asfortranarray(y.reshape(N, 1))
            row = []
            for hy in ("0", "1"):
                os.environ["SGP_HYBRID"] = hy
                ctx = L.Context(0)
                out = np.zeros(1)
                best = 1e30
                for rep in range(4):
                    t0 = time.perf_counter()
                    L.check(ctx.lib.sgp_logpdf(ctx.handle, spec.ref(), L.dptr(m), nk, L.dptr(nbuf), L.dptr(Y), N, 1, L.dptr(out)))
                    best = min(best, time.perf_counter() - t0)
                row.append((ctx.factor_schedule(N), best * 1e3, out[0]))
                ctx.close()
            assert row[0][2] == row[1][2], row
            print(f"{kind} N={N}: {row[0][0]} {row[0][1]:.2f} ms | {row[1][0]} {row[1][1]:.2f} ms | ratio {row[1][1] / row[0][1]:.3f} | logpdf bit-equal", flush=True)


if __name__ == "__main__":
    main()
