"""Small-N aggregate throughput (round-6 item 4): B independent logpdf evaluations (Matern-5/2, D = 8, different
hyper-parameters and inputs per member)
  (a) one after the other on one context                      -- the single-call latency,
  (b) on T host threads with a context each (concurrent launches on one device),
  (c) through ONE sgp_logpdf_batch call (one task pool of the dataflow kernel).
Timed around the C-ABI calls with prebuilt specs (host-side spec construction is not in the timed region).
usage: python tools/gpu_batch_time.py [N ...]      -> JSON on stdout"""
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

P = entry.load_package()
L = P.lib
PEAK = 78.6
Ns = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]
D = 8
BMAX = 16


def members(N, B, seed=0):
    rng = np.random.default_rng(seed + N)
    out = []
    for b in range(B):
        ell, s2 = 0.8 + 0.4 * rng.random(), 0.05 + 0.1 * rng.random()
        f = P.atomic(P.GP(P.with_lengthscale(P.Matern52Kernel(), ell)), P.GPC())
        x = np.asfortranarray(rng.standard_normal((D, N)))
        spec = P.build_spec(f, P.ColVecs(x))[0]
        out.append(dict(spec=spec, y=np.ascontiguousarray(rng.standard_normal(N)), nz=np.array([s2])))
    return out


def single(ctx, m, out):
    L.check(ctx.lib.sgp_logpdf(ctx.handle, m["spec"].ref(), None, L.NOISE_SCALAR, L.dptr(m["nz"]), L.dptr(m["y"]), len(m["y"]), 1,
                               L.dptr(out)))


def batch_args(ms):
    nb = len(ms)
    return ((C.POINTER(L.sgp_cov_spec) * nb)(*[C.pointer(m["spec"].c) for m in ms]),
            (C.POINTER(C.c_double) * nb)(*[L.dptr(None) for _ in ms]),
            (C.POINTER(C.c_double) * nb)(*[L.dptr(m["nz"]) for m in ms]),
            (C.POINTER(C.c_double) * nb)(*[L.dptr(m["y"]) for m in ms]))


def med(f, reps=7, warm=2):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


res = {}
for N in Ns:
    fl = N ** 3 / 3.0 + N ** 2
    ms = members(N, BMAX)
    ctx = L.Context(0)
    o1 = np.zeros(1)
    ref = []
    for m in ms:
        single(ctx, m, o1)
        ref.append(o1[0])
    t1 = med(lambda: single(ctx, ms[0], o1))
    r = dict(single_ms=t1, single_frac=fl / (t1 * 1e-3) / 1e12 / PEAK, schedule=ctx.factor_schedule(N), batch={}, threads={})
    for B in (2, 4, 8, 16):
        specs, means, noises, ys = batch_args(ms[:B])
        out = np.zeros(B)
        infos = np.zeros(B, dtype=np.int32)

        def call():
            L.check(ctx.lib.sgp_logpdf_batch(ctx.handle, B, specs, means, L.NOISE_SCALAR, noises, ys, L.dptr(out),
                                             infos.ctypes.data_as(C.POINTER(C.c_int))))
        tb = med(call)
        r["batch"][B] = dict(ms=tb, per_member_ms=tb / B, frac=B * fl / (tb * 1e-3) / 1e12 / PEAK,
                             bit_equal=bool(np.array_equal(out, np.array(ref[:B]))))
    # (b) T threads, a context each, every thread runs its own members one after the other
    for T in (2, 4, 8):
        ctxs = [L.Context(0) for _ in range(T)]
        outs = [np.zeros(1) for _ in range(T)]
        reps = 6

        def work(t):
            for _ in range(reps):
                single(ctxs[t], ms[t], outs[t])
        for t in range(T):
            single(ctxs[t], ms[t], outs[t])          # warm-up (allocations)
        th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dt = (time.perf_counter() - t0) * 1e3
        r["threads"][T] = dict(ms_per_round=dt / reps, frac=T * reps * fl / (dt * 1e-3) / 1e12 / PEAK,
                               bit_equal=bool(all(outs[t][0] == ref[t] for t in range(T))))
        for c in ctxs:
            c.close()
    ctx.close()
    res[N] = r
    print(N, json.dumps(r), file=sys.stderr)
print(json.dumps(res))
