import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgplib", os.path.join(ROOT, "stheno.jl_amd", "lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
ctx = L.Context(0)
tf = C.c_double(); err = C.c_double()
names = {0: "full", 16: "no global/LDS-store", 32: "no LDS reads", 48: "no barrier", 64: "MFMA+barrier only"}
for rep in range(2):
    for abl in (0, 16, 32, 48, 64):
        L.check(ctx.lib.sgp_bench_gemm(ctx.handle, 16384, 2048, 512, abl, 4, C.byref(tf), C.byref(err)), "gemm")
        print(f"gemm 16384x2048x512 dense, 8 waves, {names[abl]}: {tf.value:.2f} TF/s", flush=True)
