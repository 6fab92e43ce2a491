#!/bin/bash
# round 6, call 6: GPU suite after the chol_bordered split; dataflow statistics at c2 / n4k / n8k; then the round's collection (first pass)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call6
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" $OUT/pytest_gpu.log | tail -5
cd /tmp && export TMPDIR=/tmp
for c in c2 n4k; do
  SGP_DF_STATS=1 timeout 300 python $R/bench.py --config $c --steps 2 --warmup 1 --cpu-sample 0 --no-host-api --no-extras > $OUT/dfstats_$c.json 2> $OUT/dfstats_$c.err
  grep -E "dataflow n_pad|per workgroup|chain per column" $OUT/dfstats_$c.err | tail -3
done
cd $R
bash tools/collect_r06.sh > $OUT/collect.log 2>&1; tail -5 $OUT/collect.log
