"""Two contexts on one device from two host threads, both factoring under the HYBRID schedule (N = 24576 + 128 rows of the
three-block sum model, so every call launches twelve panel kernels beside its update launches): results bit-equal to the serial
run, and how many operators had to be rerun on the launches (sgp_bench_df_fallbacks).  Run on the GPU box."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
P = entry.load_package()
from stheno_jl_amd import finite_gp as fg  # noqa: E402
from stheno_jl_amd import lib as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24700
rng = np.random.default_rng(5)
F = P.gppp_sum_model()
n1 = N // 3
xs = [np.asfortranarray(rng.standard_normal((4, n))) for n in (n1, n1, N - 2 * n1)]
x = P.BlockData([P.GPPPInput(k, P.ColVecs(v)) for k, v in zip(("f1", "f2", "f3"), xs)])
y = rng.standard_normal(N)
spec, m, kind, nbuf = fg._spec_mean_noise(F(x, 0.1))
m = np.ascontiguousarray(m, dtype=np.float64)
Y = np.asfortranarray(y.reshape(N, 1))


def call(ctx):
    out = np.zeros(1)
    rc = ctx.lib.sgp_logpdf(ctx.handle, spec.ref(), L.dptr(m), kind, L.dptr(nbuf), L.dptr(Y), N, 1, L.dptr(out))
    return rc, out[0]


a, b = L.Context(0), L.Context(0)
print("schedule:", a.factor_schedule(N), flush=True)
t0 = time.perf_counter()
ref = call(a)
print("serial:", ref, f"{1e3 * (time.perf_counter() - t0):.1f} ms (first call)", flush=True)
assert ref[0] == 0 and call(b) == ref
res = {}


def worker(name, ctx):
    res[name] = [call(ctx) for _ in range(6)]


th = [threading.Thread(target=worker, args=("a", a)), threading.Thread(target=worker, args=("b", b)),
      threading.Thread(target=worker, args=("a2", a))]
t0 = time.perf_counter()
for t in th:
    t.start()
for t in th:
    t.join(300)
assert not any(t.is_alive() for t in th), "stuck"
print(f"18 concurrent calls in {time.perf_counter() - t0:.2f} s", flush=True)
for k, v in res.items():
    assert all(r == ref for r in v), (k, v, ref)
for name, ctx in (("a", a), ("b", b)):
    fb = C.c_int64()
    L.check(ctx.bench.sgp_bench_df_fallbacks(ctx.handle, C.byref(fb)))
    print("context", name, "operators rerun on the launches after a wait bound:", fb.value)
print("bit-equal to the serial run: ok")
