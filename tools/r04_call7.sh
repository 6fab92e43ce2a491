#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04g; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 ) > $O/pytest_gpu.txt
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04g/bench_default.json'))
print('c5', round(d['ms_per_step'],1), 'frac', round(d['roofline']['frac'],4), 'parity', d['parity_rel'])
ns=d['north_star_target']; print('target', round(ns['ms_per_step'],1), round(ns['frac'],4), ns['parity_rel'])
for k,v in d['sizes'].items(): print(k, round(v['ms_per_step'],3), round(v['frac'],4), v['parity_rel'], v['schedule'])
PY
# tall bordered VFE shape: dataflow vs launches (M = 4096, N = 60000, unchunked)
python - <<'PY'
import os, time, numpy as np, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import __graft_entry__ as g
P = g.load_package()
rng = np.random.default_rng(3)
N, M, D = 60000, 4096, 8
X = np.asfortranarray(rng.standard_normal((D, N))); Z = np.asfortranarray(X[:, rng.permutation(N)[:M]]); y = rng.standard_normal(N)
f = P.stretch(P.atomic(P.GP(P.SEKernel()), P.GPC()), 1 / np.sqrt(D))
for df in ("0", "1"):
    os.environ["SGP_DATAFLOW"] = df; os.environ["SGP_VFE_CHUNK"] = "0"
    ctx = P.lib.Context(0); prev = P.lib.set_default_context(ctx)
    ts = []
    for r in range(4):
        t0 = time.perf_counter(); e = P.elbo(P.VFE(f(P.ColVecs(Z), 1e-6)), f(P.ColVecs(X), 0.1), y); ts.append((time.perf_counter() - t0) * 1e3)
    print("ELBO M=4096 N=60000 unchunked SGP_DATAFLOW=" + df, "best %.1f ms" % min(ts[1:]), repr(e))
    P.lib.set_default_context(prev); ctx.close()
PY
