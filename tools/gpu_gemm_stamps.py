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
span = drained.max()
# s_memtime rate: calibrate against the known MFMA time of the contraction is circular; report ticks and ratios
pro, body, epi = land - ent, done - land, drained - done
tmap = d[:, 7] - d[:, 0]
print(f"  id -> tile mapping (tile_of_id)                      mean {tmap.mean():9.0f}  p50 {np.percentile(tmap, 50):9.0f}  p90 {np.percentile(tmap, 90):9.0f}")
print(f"lower {m}^2 K={k}: {len(d)} tiles, launch span {span} ticks")
for name, v in (("prologue (entry -> first chunk + C tile landed)", pro), ("contraction", body), ("epilogue (stores drained)", epi)):
    print(f"  {name:52s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f}  max {v.max():9.0f}")
tile = drained - ent
print(f"  whole tile                                           mean {tile.mean():9.0f}; prologue + epilogue = {100 * (pro.mean() + epi.mean()) / tile.mean():.1f} % of a tile's residence")
# per-slot timelines: workgroups that ran on the same CU (XCC, SE, SH, CU from HW_ID) in start order; the gap between one
# workgroup's drain and the next entry on that CU, given two slots per CU
cu = d[:, 4] & ~0xf0          # drop the wave-slot / SIMD bits of HW_ID (bits 7:4 = SIMD, 3:0 = wave)
gaps, busy = [], []
for key in np.unique(cu):
    sel = np.argsort(ent[cu == key])
    e, x = ent[cu == key][sel], drained[cu == key][sel]
    # two slots: greedily assign each workgroup to the slot that freed first
    free = [0, 0]
    for a, b in zip(e, x):
        s = 0 if free[0] <= free[1] else 1
        if free[s] > 0:
            gaps.append(a - free[s])
        free[s] = b
    busy.append((x - e).sum() / (2.0 * span))
gaps = np.array(gaps)
print(f"  CUs seen {len(np.unique(cu))}; slot re-use gap (previous workgroup drained -> next entry): mean {gaps.mean():.0f} p50 {np.percentile(gaps, 50):.0f} "
      f"p90 {np.percentile(gaps, 90):.0f} ticks = {100 * gaps.mean() / tile.mean():.1f} % of a tile; slot occupancy {np.mean(busy):.3f}")
ideal = 2.0 * 128 * 128 * k / (78.6e12 / 256 / 2)   # seconds per tile at half a CU's share of the datasheet peak
print(f"  (a tile's contraction at half a CU of the 78.6 TFLOP/s peak: {ideal * 1e6:.1f} us; contraction mean / that = s_memtime ticks per us x eff)")
