"""Where a trailing-update tile's time goes: per-workgroup phase stamps of the production tile program
(sgp_bench_gemm_stamps) for one isolated lower update, reduced to per-slot timelines.
usage: gpu_gemm_stamps.py [m=32768] [k=1024]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

P = g.load_package()
L = P.lib
m = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ctx = L.Context(0)
n = C.c_int64()
L.check(ctx.lib.sgp_bench_gemm_stamps(ctx.handle, m, k, None, 0, C.byref(n)))
buf = np.zeros(8 * n.value, dtype=np.int64)
L.check(ctx.lib.sgp_bench_gemm_stamps(ctx.handle, m, k, buf.ctypes.data_as(C.POINTER(C.c_longlong)), buf.size, C.byref(n)))
d = buf.reshape(-1, 8)
d = d[d[:, 0] > 0]
t0 = d[:, 0].min()
ent, land, done, drained = (d[:, i] - t0 for i in range(4))
# s_memtime rate: calibrate against the known MFMA time of the contraction is circular; report ticks and ratios
pro, body, epi = land - ent, done - land, drained - done
mode = ("  [tiles dealt at random" + (", k offset per workgroup" if os.environ.get("SGP_STAMP_ROTATE") else "") + "]") \
    if os.environ.get("SGP_STAMP_SCRAMBLE") else ""
print(f"launch span (first entry -> last store drained): {int(d[:, 3].max() - d[:, 0].min())} ticks{mode}")
tmap = d[:, 7] - d[:, 0]
print(f"  id -> tile mapping (tile_of_id)                      mean {tmap.mean():9.0f}  p50 {np.percentile(tmap, 50):9.0f}  p90 {np.percentile(tmap, 90):9.0f}")
print(f"lower {m}^2 K={k}: {len(d)} tiles (s_memtime ticks = shader-clock cycles)")
for name, v in (("prologue (entry -> first chunk + C tile landed)", pro), ("contraction", body), ("epilogue (stores drained)", epi)):
    print(f"  {name:52s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f}")
tile = drained - ent
print(f"  whole tile                                           mean {tile.mean():9.0f}; prologue + epilogue = {100 * (pro.mean() + epi.mean()) / tile.mean():.1f} % of a tile's residence")
ideal = 2.0 * 128 * 128 * k / (78.6e12 / 256 / 2)   # seconds per tile at half a CU's share of the datasheet peak
print(f"  (a tile's contraction at half a CU of the 78.6 TFLOP/s peak: {ideal * 1e6:.1f} us; contraction mean / that = s_memtime ticks per us x eff)")
