#!/bin/bash
# round 6, call 5: full GPU suite after the prune + the library split + the sharded covariance / input-gradient operators
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call5
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -15
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 900 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -2 $OUT/bench_default.err
python - <<'PY'
import json,os
d=json.load(open(os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r06_call5/bench_default.json"))
print({k:d[k] for k in ("value","ms_per_step","parity_rel")})
print({k:{kk:v.get(kk) for kk in ("ms_per_step","ms_per_call","frac","every_member_bit_equal_to_its_own_call","member0_parity_rel")} for k,v in d["sizes"].items()})
PY
