"""logpdf beyond the BASELINE sizes: does the single-GPU path hold its rate when the factor matrix takes a large part
of the 288 GB (N = 98 304: 77 GB, N = 131 072: 137 GB)?  Device-resident entry point, Matern-5/2, D = 8."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
import bench_configs as bc
P = g.load_package(); L = P.lib
ctx = L.default_context(); lib = ctx.lib
for N in [int(a) for a in sys.argv[1:]] or [98304, 131072]:
    D = 8
    X, y = bc.make_inputs(N, D)
    f = P.stretch(P.atomic(P.GP(P.Matern52Kernel()), P.GPC()), 1.0 / np.sqrt(D))
    spec = P.build_spec(f, P.ColVecs(X))[0]
    ds = C.c_void_p(); L.check(lib.sgp_dspec_create(ctx.handle, spec.ref(), C.byref(ds)), "dspec")
    npad, mtot = C.c_int64(), C.c_int64(); lib.sgp_geometry(N, 1, C.byref(npad), C.byref(mtot))
    A = torch.empty(npad.value * mtot.value, dtype=torch.float64, device="cuda")
    dY = torch.from_numpy(y).cuda(); out = np.zeros(1); nz = np.array([0.1]); tm = np.zeros(8)
    for it in range(2):
        t0 = time.perf_counter()
        L.check(lib.sgp_dev_logpdf(ctx.handle, ds, A.data_ptr(), None, L.NOISE_SCALAR, L.dptr(nz), None, dY.data_ptr(), N, 1,
                                   L.dptr(out), L.dptr(tm)), "logpdf")
        dt = time.perf_counter() - t0
    print(f"N={N}: {1e3*dt:.0f} ms, {N**3/3/dt/1e12:.1f} TFLOP/s whole step ({N**3/3/dt/1e12/78.6:.3f} of peak), matrix "
          f"{8*npad.value*mtot.value/1e9:.0f} GB, assembly {tm[0]:.1f} ms, trailing updates {tm[5]/(tm[3]*1e-3)/1e12:.1f} TFLOP/s per launch, "
          f"logpdf {out[0]:.6f}", flush=True)
    lib.sgp_dspec_destroy(ds); del A; torch.cuda.empty_cache()
