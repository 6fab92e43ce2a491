import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgplib", os.path.join(ROOT, "stheno.jl_amd", "lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
ctx = L.Context(0)
tf = C.c_double(); err = C.c_double()
variants = {0: "DMA kernel (production)", 2: "register-staged baseline"}
for rep in range(2):
    for (m, n, k) in [(16384, 2048, 512), (16384, 16384, 512), (16384, 16384, 128), (32768, 32768, 1024)]:
        for flag, name in variants.items():
            lo = 1 if n == m else 0
            L.check(ctx.lib.sgp_bench_gemm(ctx.handle, m, n, k, lo | flag, 3, C.byref(tf), C.byref(err)), "gemm")
            print(f"gemm {m}x{n}x{k} lower={lo} [{name}]: {tf.value:.2f} TF/s err {err.value:.1e}", flush=True)
