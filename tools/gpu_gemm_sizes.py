"""Trailing-update GEMM rate against problem size and depth (isolated launches):
C(m x n lower) -= P P', for the shapes the factorisation actually issues."""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgplib", os.path.join(ROOT, "stheno.jl_amd", "lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
ctx = L.Context(0)
tf = C.c_double(); err = C.c_double()
for m, k in [(8192, 512), (16384, 512), (16384, 1024), (32768, 1024), (49152, 1024), (65536, 1024), (65536, 2048)]:
    L.check(ctx.lib.sgp_bench_gemm(ctx.handle, m, m, k, 1, 3, C.byref(tf), C.byref(err)), "gemm")
    print(f"lower {m}^2 K={k}: {tf.value:.2f} TF/s ({tf.value/78.6:.3f} of 78.6)", flush=True)
