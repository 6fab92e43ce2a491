"""Phase breakdown of sgp::potrf_diag_kernel from s_memtime stamps of wave 0 (sgp_bench_potrf)."""
import ctypes as C, importlib.util, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgplib", os.path.join(ROOT, "stheno.jl_amd", "lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
ctx = L.Context(0)
us = C.c_double(); st = (C.c_longlong * 64)()
L.check(ctx.lib.sgp_bench_potrf(ctx.handle, 20, C.byref(us), st), "potrf")
v = [x for x in st if x]
d = [b - a for a, b in zip(v, v[1:])]
print(f"potrf_diag launch (HIP events): {us.value:.1f} us; s_memtime ticks start->end: {v[-1]-v[0]} (100 MHz => {(v[-1]-v[0])/100:.1f} us)")
names = ["load tile + sync", "microchol(0)", "sync"]
for cb in range(8):
    names += [f"solve({cb})", "sync"]
    if cb < 7:
        names += [f"trailing update({cb}) [wave 0: 1 block]", f"microchol({cb+1})", "sync"]
names += ["(loop exit)", "store tile", "logdet + exit"]
for n, x in zip(names, d):
    print(f"  {n:40s} {x:6d} ticks")
