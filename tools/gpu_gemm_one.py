"""One representative trailing-update launch shape for PMC passes: C(32768^2 lower) -= P P', K = 1024."""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgplib", os.path.join(ROOT, "stheno.jl_amd", "lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
ctx = L.Context(0)
tf = C.c_double(); err = C.c_double()
m, k = 32768, 1024
L.check(ctx.lib.sgp_bench_gemm(ctx.handle, m, m, k, 1, 3, C.byref(tf), C.byref(err)), "gemm")
flops = k * m * (m + 1.0)
print(f"gemm {m}x{m}x{k} lower: {tf.value:.2f} TF/s, algorithmic flops/launch {flops:.4g}, "
      f"algorithmic bytes/launch (C r+w, panel once) {8.0*m*(m+1)/2*2 + 8.0*m*k:.4g}", flush=True)
