#!/bin/bash
# Round 5: hybrid with both update launches on one stream (column update first), the panel kernel on the other
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05v
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() {
  c=$1; st=$2; shift 2
  env "$@" timeout 300 python $R/bench.py --config $c --steps $st --warmup 1 --cpu-sample 0 --no-host-api --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c', '$*', round(d['ms_per_step'],3), 'ms', repr(d['logpdf']), r['schedule'][:12], 'frac', round(r['frac'],4), 'launches', r.get('launches'), 'avg', round(r['avg_launch_ms'],3), 'busy', round(r.get('busy_ms') or 0,1))"
}
{
run c5 3 X=default
run c5 3 SGP_HYBRID_ORDER=1
run target 3 X=default
run target 3 SGP_HYBRID_ORDER=1
run n32k 5 X=default
run n32k 5 SGP_HYBRID_ORDER=1
run c3 5 X=default
run c3 5 SGP_HYBRID_ORDER=1
} 2>&1 | tee $OUT/hybrid_order.txt
cd $R
SGP_HYBRID_ORDER=1 timeout 300 python -m pytest tests/test_gpu_dataflow.py -m gpu -q -x -p no:cacheprovider -k "hybrid" > $OUT/pytest_order.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_order.log
