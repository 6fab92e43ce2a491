#!/bin/bash
# round 6, call 4: full GPU suite, the mid-N schedule sweep, the default bench line with its new legs
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call4
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|rc=" $OUT/pytest_gpu.log | tail -3
timeout 900 python tools/gpu_midn_sweep.py 12288 16384 20480 > $OUT/midn_sweep.json 2> $OUT/midn_sweep.err; tail -3 $OUT/midn_sweep.err; grep -v "^{" $OUT/midn_sweep.json
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -2 $OUT/bench_default.err
python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r06_call4/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","parity_rel")})
r=d["roofline"]; print({k:r.get(k) for k in ("frac","achieved_while_busy","round4_schedule_same_box")})
print("target",d["north_star_target"]["ms_per_step"]); print({k:v["ms_per_step"] for k,v in d["sizes"].items()})
PY
