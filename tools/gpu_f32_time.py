"""fp32 logpdf (sgp_logpdf_f32) against the fp64 path at a few sizes: time and agreement."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import bench_configs as bc
P = g.load_package()
for N in [int(a) for a in sys.argv[1:]] or [4096, 16384, 32768]:
    D = 8
    X, y = bc.make_inputs(N, D)
    f = P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 1.0 / np.sqrt(D))
    fx64 = f(P.ColVecs(X), 0.1)
    fx32 = f(P.ColVecs(X.astype(np.float32)), np.float32(0.1))
    y32 = y.astype(np.float32)
    v64 = P.logpdf(fx64, y); v32 = P.logpdf(fx32, y32)
    t0 = time.perf_counter(); v64 = P.logpdf(fx64, y); t1 = time.perf_counter(); v32 = P.logpdf(fx32, y32); t2 = time.perf_counter()
    print(f"N={N}: fp64 {1e3*(t1-t0):.2f} ms ({N**3/3/(t1-t0)/1e12:.1f} TFLOP/s), fp32 {1e3*(t2-t1):.2f} ms "
          f"({N**3/3/(t2-t1)/1e12:.1f} TFLOP/s), values {v64:.6f} / {float(v32):.3f}, rel {abs(float(v32)-v64)/abs(v64):.2e}", flush=True)
