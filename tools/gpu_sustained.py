import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgplib", os.path.join(ROOT, "stheno.jl_amd", "lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
ctx = L.Context(0)
tf = C.c_double(); err = C.c_double()
L.check(ctx.lib.sgp_bench_mfma_f64(ctx.handle, int(sys.argv[1]), C.byref(tf), C.byref(err)), "mfma")
