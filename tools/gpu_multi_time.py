"""One-process multi-rank context (sgp_ctx_create_multi) on the single GPU of the box: the same logpdf
through 1 rank over RCCL / peer copies and 2 / 4 loopback ranks on device 0, against the CPU golden."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402
import bench_configs as bc  # noqa: E402

P = g.load_package()
for name in sys.argv[1:] or ["c2", "c5"]:
    w = bc.build(P, name)
    gold = bc.golden(name)["logpdf"]
    for devs, env in (([0], "rccl"), ([0], "p2p"), ([0, 0], "auto"), ([0, 0, 0, 0], "auto")):
        os.environ["SGP_MULTI_TRANSPORT"] = env
        ctx = P.lib.Context(devices=devs)
        prev = P.lib.set_default_context(ctx)
        try:
            v = P.logpdf(w["fx"], w["y"])
            n = 3 if w["N"] <= 16384 else 1
            t0 = time.perf_counter()
            for _ in range(n):
                v = P.logpdf(w["fx"], w["y"])
            ms = (time.perf_counter() - t0) / n * 1e3
        finally:
            P.lib.set_default_context(prev)
        print(f"multi ctx {name} ranks={len(devs)} transport={ctx.transport}: {ms:.1f} ms/logpdf (host mirror incl. "
              f"flatten + uploads), |rel| vs CPU golden {abs(v - gold) / abs(gold):.2e}", flush=True)
        ctx.close()
