"""logpdf vs logpdf + gradient at one size (single GP, Matern-5/2, D = 8), 3 timed calls each; under rocprofv3 --kernel-trace
--stats the per-kernel totals give the stage split of the gradient.  usage: python tools/gpu_grad_split.py N"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
P = entry.load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
rng = np.random.default_rng(0)
D = 8
X = P.ColVecs(rng.standard_normal((D, N)) / np.sqrt(D))
y = rng.standard_normal(N)
f = P.atomic(P.GP(P.Matern52Kernel()), P.GPC())
fx = f(X, 0.1)
P.logpdf(fx, y); P.logpdf_and_gradient(fx, y)
t0 = time.perf_counter()
for _ in range(3):
    P.logpdf(fx, y)
t1 = time.perf_counter()
for _ in range(3):
    g = P.logpdf_and_gradient(fx, y)
t2 = time.perf_counter()
print(f"N={N}: logpdf {1e3 * (t1 - t0) / 3:.2f} ms, logpdf+grad {1e3 * (t2 - t1) / 3:.2f} ms, ratio {(t2 - t1) / (t1 - t0):.2f}, "
      f"N^3 rate {N ** 3 / ((t2 - t1) / 3) / 1e12:.1f} TFLOP/s", flush=True)
