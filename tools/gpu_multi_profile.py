"""Serialised per-panel timings of the sharded factorisation with P loopback ranks on ONE GPU (sgp_ctx_multi_profile):
what each GPU of a P-GPU node would spend on its own share -- input of tools/multi_projection.py.
usage: python tools/gpu_multi_profile.py <config> <P> <out.json> [panel]   (SGP_MULTI_GROUP / _PANEL_TAIL / _TAIL_FRAC from the
environment)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
import bench_configs as bc  # noqa: E402

cfg, P, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
if len(sys.argv) > 4:
    os.environ["SGP_MULTI_PANEL"] = sys.argv[4]
pkg = entry.load_package()
L = pkg.lib
kind, N, D = bc.CONFIGS[cfg]
w = bc.build(pkg, cfg)
spec = pkg.build_spec(w["f"], w["x"])[0]
y = np.ascontiguousarray(w["y"])
ctx = L.Context(devices=[0] * P)
res = np.zeros(1)
nz = np.array([bc.SIGMA2])


def call():
    t0 = time.perf_counter()
    L.check(ctx.lib.sgp_logpdf(ctx.handle, spec.ref(), None, L.NOISE_SCALAR, L.dptr(nz), L.dptr(y), N, 1, L.dptr(res)))
    return (time.perf_counter() - t0) * 1e3


call()                       # warm-up (allocations)
overlapped_ms = call()       # the normal (overlapped) schedule with P ranks sharing the one GPU
st = np.zeros(11 + 4 * P)
n = C.c_int64()
L.check(ctx.lib.sgp_ctx_multi_stats(ctx.handle, L.dptr(st), len(st), C.byref(n)))
L.check(ctx.lib.sgp_ctx_multi_owners(ctx.handle, None, 0, C.byref(n)))
own = (C.c_int32 * n.value)()
L.check(ctx.lib.sgp_ctx_multi_owners(ctx.handle, own, n.value, C.byref(n)))
owners = [int(v) for v in own]
L.check(ctx.lib.sgp_ctx_multi_profile(ctx.handle, 1))
serial_ms = call()
L.check(ctx.lib.sgp_ctx_multi_profile_get(ctx.handle, None, 0, C.byref(n)))
prof = np.zeros(n.value)
L.check(ctx.lib.sgp_ctx_multi_profile_get(ctx.handle, L.dptr(prof), n.value, C.byref(n)))
prof = prof.reshape(-1, 3 + 3 * P)
L.check(ctx.bench.sgp_bench_multi_profile_pieces(ctx.handle, None, 0, C.byref(n)))
pcs = np.zeros(n.value)
L.check(ctx.bench.sgp_bench_multi_profile_pieces(ctx.handle, L.dptr(pcs), n.value, C.byref(n)))
pcs = pcs.reshape(-1, 8)
g = bc.golden(cfg)
n_pad = (N + 127) // 128 * 128
m_tot = n_pad + 128
sub = int(os.environ.get("SGP_MULTI_SUBPANEL", "512")) // 128 * 128   # (the library's default, multi.hip: sgp_multi::sub; round 4
# recorded 256 here while the library ran 512: its projections modelled a 4-deep pipeline where 2 pieces travelled)
widths, c0 = [], 0
for b in prof[:, 2]:                       # panel bytes = 8 * (m_tot - col0) * width
    w = int(round(b / 8.0 / (m_tot - c0)))
    widths.append(w)
    c0 += w
nsub = [min(8, -(-w // sub)) if sub >= 128 else 1 for w in widths]
pieces = []      # (round 6 measured uneven pieces through a library switch that is gone: profiles/r06_experiments/sharded_chain.md)
piece_frac = None
if pieces:       # uneven pieces (round 6): panels whose width the pieces add up to
    nsub = [len(pieces) if w == sum(pieces) else n for w, n in zip(widths, nsub)]
    piece_frac = [p / sum(pieces) for p in pieces]
json.dump({"config": cfg, "N": N, "ranks": P, "panel_width": int(st[4]), "panels": int(st[5]), "group": int(st[7]),
           "subpanel": sub, "widths": widths, "nsub": nsub, "owners": owners, "pieces": pieces, "piece_frac": piece_frac,
           "piece_ms": pcs.tolist(),      # per panel: the ms of each sub-panel's launches (8 slots)
           # round 6: panels factored by launches of the dataflow kernel; the look-ahead update with the previous panel's LAST
           # sub-panel rides in the first of them (then factor_ms holds it and lookahead_update_ms only the earlier pieces)
           "panel_df": int(os.environ.get("SGP_MULTI_PANEL_DF", "1")) if os.environ.get("SGP_HYBRID", "") != "0" else 0,
           "fuse_la": int(os.environ.get("SGP_MULTI_FUSE_LA", "1")) if (os.environ.get("SGP_HYBRID", "") != "0" and
                                                                       int(os.environ.get("SGP_MULTI_PANEL_DF", "1"))) else 0,
           "ownership": {0: "cyclic", 1: "balanced", 2: "list"}.get(int(st[9 + 4 * P]), "?"),
           "layout": "per panel: factor_ms, lookahead_update_ms, panel_bytes, then per rank near_a_ms, near_b_ms, far_ms",
           "logpdf": float(res[0]), "parity_rel": None if g is None else abs(res[0] - g["logpdf"]) / abs(g["logpdf"]),
           "one_gpu_overlapped_ms": overlapped_ms, "one_gpu_serialised_ms": serial_ms, "host_enqueue_ms": float(st[8 + 4 * P]),
           "columns": ["factor_ms", "lookahead_update_ms", "panel_bytes"] +
                      [f"{c}_ms_rank{i}" for i in range(P) for c in ("near_a", "near_b", "far")],
           "per_panel": prof.tolist()}, open(out, "w"))
print(cfg, "P", P, "host enqueue", round(float(st[8 + 4 * P]), 1), "overlapped", round(overlapped_ms, 1), "serialised", round(serial_ms, 1), "factor sum", round(prof[:, 0].sum(), 1),
      "la sum", round(prof[:, 1].sum(), 1), "update sum per rank (near a + near b + far)",
      np.round(prof[:, 3:].reshape(len(prof), P, 3).sum(axis=(0, 2)), 1).tolist())
