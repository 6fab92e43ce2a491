"""GPU bring-up 2: MFMA variants, GEMM rates, first resident-logpdf timings.
usage: python tools/gpu_trip2.py [mfma|gemm|logpdf N ...]"""
import ctypes as C
import importlib.util
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("sgplib", os.path.join(ROOT, "stheno.jl_amd", "lib.py"))
L = importlib.util.module_from_spec(spec)
spec.loader.exec_module(L)
ctx = L.Context(0)
lib = ctx.lib
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
tf = C.c_double()
err = C.c_double()

if mode in ("mfma", "all"):
    L.check(lib.sgp_bench_mfma_f64(ctx.handle, 20000, C.byref(tf), C.byref(err)), "mfma")
    print("mfma best", tf.value, "layout err", err.value, flush=True)

if mode in ("gemm", "all"):
    for (m, n, k, lo) in [(16384, 16384, 512, 1), (16384, 16384, 128, 1), (16384, 2048, 512, 0)]:
        L.check(lib.sgp_bench_gemm(ctx.handle, m, n, k, lo, 5, C.byref(tf), C.byref(err)), "gemm")
        print("gemm", m, n, k, lo, "TF/s", tf.value, "maxerr", err.value, flush=True)

if mode in ("logpdf", "all"):
    sizes = [int(a) for a in sys.argv[2:]] or [16384]
    for N in sizes:
        D = 8
        rng = np.random.default_rng(123456)
        X = np.asfortranarray(rng.standard_normal((D, N)) / np.sqrt(D))
        y = rng.standard_normal(N)
        sp = L.Spec([N], [N], [X], {(0, 0): [(L.MATERN52, 0, 0, 1.0, 0.0, None, None)]}, True)
        ds = C.c_void_p()
        L.check(lib.sgp_dspec_create(ctx.handle, sp.ref(), C.byref(ds)), "dspec")
        npad = C.c_int64()
        mtot = C.c_int64()
        lib.sgp_geometry(N, 1, C.byref(npad), C.byref(mtot))
        A = torch.empty(npad.value * mtot.value, dtype=torch.float64, device="cuda")
        dY = torch.from_numpy(y).cuda()
        o = np.zeros(1)
        tm = np.zeros(8)
        nz = np.array([0.1])
        for it in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            L.check(lib.sgp_dev_logpdf(ctx.handle, ds, A.data_ptr(), None, L.NOISE_SCALAR, L.dptr(nz), None,
                                       dY.data_ptr(), N, 1, L.dptr(o), L.dptr(tm)), "dev_logpdf")
            dt = time.time() - t0
            chol_tf = N ** 3 / 3 / (tm[1] * 1e-3) / 1e12
            upd_tf = tm[5] / (tm[3] * 1e-3) / 1e12 if tm[3] > 0 else 0
            print(f"N={N} logpdf={o[0]:.6f} wall={dt*1e3:.1f}ms assemble={tm[0]:.2f}ms chol={tm[1]:.2f}ms "
                  f"({chol_tf:.1f} TF/s) final={tm[2]:.2f}ms updates={tm[3]:.2f}ms over {int(tm[4])} launches "
                  f"({upd_tf:.1f} TF/s)", flush=True)
        lib.sgp_dspec_destroy(ds)
        del A
print("TRIP2 DONE")
