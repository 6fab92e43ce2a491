"""fp32 logpdf (sgp_logpdf_f32) under driver variants (environment: SGP_F32_SERIAL_N / _WOUT / _WMID, read once per process):
usage: gpu_f32_variants.py N [N ...]  -- prints best-of-3 ms and the whole-step TFLOP/s."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
import bench_configs as bc
P = g.load_package()
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("SGP_F32"))
for N in [int(a) for a in sys.argv[1:]]:
    D = 8
    X, y = bc.make_inputs(N, D)
    f = P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 1.0 / np.sqrt(D))
    fx32 = f(P.ColVecs(X.astype(np.float32)), np.float32(0.1))
    y32 = y.astype(np.float32)
    ts = []
    for r in range(4):
        t0 = time.perf_counter(); v32 = P.logpdf(fx32, y32); ts.append(time.perf_counter() - t0)
    b = min(ts[1:])
    print(f"[{tag or 'default'}] N={N}: fp32 {1e3*b:.2f} ms ({N**3/3/b/1e12:.1f} TFLOP/s = {N**3/3/b/1e12/157.3:.3f} of peak), value {float(v32):.3f}", flush=True)
