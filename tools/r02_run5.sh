#!/bin/bash
# GPU pass 5: small-N knob matrix + contention check at c2 + dist (packed) + multi ctx timing
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02e
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_multi.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
one() { echo -n "$1 $2 " ; env $1 timeout 300 python $R/bench.py --config $2 --steps $3 --warmup 3 --cpu-sample 0 --no-host-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['parity_rel'])"; }
for la in 0 1; do for w in 256 512 1024 2048; do for c in c1 n4k; do one "SGP_LOOKAHEAD=$la SGP_WOUT=$w" $c 30; done; done; done | tee $OUT/small_knobs.txt
for la in 0 1; do for w in 512 1024; do one "SGP_LOOKAHEAD=$la SGP_WOUT=$w" c2 20; done; done | tee -a $OUT/small_knobs.txt
SGP_LOOKAHEAD=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2_la0 -o c2 -- \
    python $R/bench.py --config c2 --steps 3 --warmup 2 --cpu-sample 0 --no-host-api > $OUT/prof_c2_la0.json 2> $OUT/prof_c2_la0.err
f=$(find $OUT/prof_c2_la0 -name "*kernel_stats.csv" | head -1); head -12 $f
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
timeout 300 python $R/bench.py --config c2 --force-dist --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dist world1 c2', d['ms_per_step'], d['parity_rel'])"
timeout 300 python $R/bench.py --config c5 --force-dist --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dist world1 c5', d['ms_per_step'], d['parity_rel'])"
python - <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
P = g.load_package()
import bench_configs as bc
import numpy as np
for name in ("c2", "c5"):
    w = bc.build(P, name)
    gold = bc.golden(name)["logpdf"]
    for devs, env in (([0], "rccl"), ([0], "p2p"), ([0, 0], "auto"), ([0, 0, 0, 0], "auto")):
        os.environ["SGP_MULTI_TRANSPORT"] = env
        ctx = P.lib.Context(devices=devs)
        prev = P.lib.set_default_context(ctx)
        try:
            v = P.logpdf(w["fx"], w["y"])
            t0 = time.perf_counter(); n = 3 if name == "c2" else 1
            for _ in range(n):
                v = P.logpdf(w["fx"], w["y"])
            ms = (time.perf_counter() - t0) / n * 1e3
        finally:
            P.lib.set_default_context(prev)
        print(f"multi ctx {name} ranks={len(devs)} transport={ctx.transport}: {ms:.1f} ms/logpdf (host mirror incl. flatten), rel {abs(v-gold)/abs(gold):.2e}", flush=True)
        ctx.close()
PY
