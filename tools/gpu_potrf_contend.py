"""potrf_diag under the look-ahead's contention (sgp_bench_potrf_contended): trailing updates run on the update
stream while potrf_diag launches go down the panel stream one by one.  Separates 'waits for a workgroup slot'
(HIP-event launch time minus the kernel's own s_memtime span) from 'runs slower once resident'."""
import ctypes as C, importlib.util, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgplib", os.path.join(ROOT, "stheno.jl_amd", "lib.py"))
L = importlib.util.module_from_spec(spec); spec.loader.exec_module(L)
ctx = L.Context(0)
def q(v, f):
    v = sorted(v); return v[min(len(v) - 1, int(len(v) * f))]
for m, k, g, n in [(32768, 1024, 12, 1500), (16384, 512, 60, 1500), (8192, 512, 200, 1500)]:
    us = (C.c_double * n)(); tk = (C.c_longlong * n)(); bz = (C.c_int * n)()
    L.check(ctx.lib.sgp_bench_potrf_contended(ctx.handle, m, k, g, n, us, tk, bz), "contended")
    busy = [(us[i], tk[i]) for i in range(n) if bz[i]]
    idle = [(us[i], tk[i]) for i in range(n) if not bz[i]][5:]
    for name, sel in (("updates running", busy), ("chip idle", idle)):
        if not sel:
            print(f"update {m}^2 K={k}: {name}: no samples"); continue
        u = [x[0] for x in sel]; t = [x[1] for x in sel]
        print(f"update {m}^2 K={k}: {name}: n={len(sel)}  launch us (events) median {q(u,.5):.1f} p90 {q(u,.9):.1f} max {max(u):.1f}"
              f" | in-kernel ticks median {q(t,.5)} p90 {q(t,.9)} max {max(t)}", flush=True)
