"""One covariance-assembly launch per iteration (lower tiles of K + s2 I for a BASELINE config), for
rocprofv3 kernel timing / --pmc passes on sgp::assemble_block2_kernel.  usage: gpu_assemble_one.py [config] [iters]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402
import bench_configs as bc  # noqa: E402

P = entry.load_package()
L = P.lib
name = sys.argv[1] if len(sys.argv) > 1 else "c5"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kind, N, D = bc.CONFIGS[name]
w = bc.build(P, name)
spec = P.build_spec(w["f"], w["x"])[0]
ctx = L.Context(0)
lib = ctx.lib
ds = C.c_void_p()
L.check(lib.sgp_dspec_create(ctx.handle, spec.ref(), C.byref(ds)))
npad, mtot = C.c_int64(), C.c_int64()
lib.sgp_geometry(N, 1, C.byref(npad), C.byref(mtot))
A = torch.empty(npad.value * mtot.value, dtype=torch.float64, device="cuda")
dY = torch.from_numpy(w["y"]).cuda()
nz = np.array([bc.SIGMA2])
st = torch.cuda.Stream()


def run():
    L.check(lib.sgp_dev_assemble_cols(ctx.handle, ds, N, 0, npad.value, A.data_ptr(), mtot.value, mtot.value, None,
                                      L.NOISE_SCALAR, L.dptr(nz), None, dY.data_ptr(), N, 1, st.cuda_stream))


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    run()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
b = 8.0 * N * (N + 1) / 2 + 8.0 * D * N
print(f"assemble {name}: {ms:.3f} ms/launch, {b / ms / 1e6:.0f} GB/s algorithmic (lower triangle written once: {b / 1e9:.2f} GB)")
