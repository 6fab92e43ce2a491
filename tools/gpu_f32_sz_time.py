"""fp32 logpdf of the structured north-star model (f3 = f1 + f2 at n points each, ordered f1, f2, f3) with the structural
zeros skipped and not (SGP_STRUCT_ZEROS): wall time per call and the bits.  Usage: python tools/gpu_f32_sz_time.py [--out FILE] [n ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

P = entry.load_package()


def main():
    out_path = "gpurun_out/f32_sz_time.json"
    args = sys.argv[1:]
    if "--out" in args:
        i = args.index("--out")
        out_path = args[i + 1]
        del args[i:i + 2]
    ns = [int(a) for a in args] or [4096, 10923, 21845]
    out = []
    for n in ns:
        rng = np.random.default_rng(n)
        xs = {k: np.asfortranarray(rng.standard_normal((2, n)).astype(np.float32)) for k in ("f1", "f2", "f3")}
        y = rng.standard_normal(3 * n).astype(np.float32)
        F = P.gppp_sum_model()
        x = P.BlockData([P.GPPPInput(k, P.ColVecs(xs[k])) for k in ("f1", "f2", "f3")])
        rec = {"n_per_block": n, "N": 3 * n}
        for sz in (0, 1):
            os.environ["SGP_STRUCT_ZEROS"] = str(sz)
            ctx = P.lib.Context(0)
            prev = P.lib.set_default_context(ctx)
            try:
                lp = P.logpdf(F(x, np.float32(0.1)), y)
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    lp = P.logpdf(F(x, np.float32(0.1)), y)
                    ts.append((time.perf_counter() - t0) * 1e3)
                e, d = ctx.factor_work()
            finally:
                P.lib.set_default_context(prev)
                ctx.close()
            rec["sz%d" % sz] = {"ms": min(ts), "logpdf": float(lp), "work": [e, d]}
        rec["bit_equal"] = rec["sz0"]["logpdf"] == rec["sz1"]["logpdf"]
        rec["speedup"] = rec["sz0"]["ms"] / rec["sz1"]["ms"]
        print(json.dumps(rec), flush=True)
        out.append(rec)
    os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
    json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
