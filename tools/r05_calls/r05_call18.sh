#!/bin/bash
# Round 5: hybrid with sub-panels inside wider outer panels (K = 4096 / 8192 trailing updates, 2048-column dataflow panels)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05u
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() {
  c=$1; st=$2; shift 2
  env "$@" timeout 300 python $R/bench.py --config $c --steps $st --warmup 1 --cpu-sample 0 --no-host-api --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c', '$*', round(d['ms_per_step'],3), 'ms', repr(d['logpdf']), r['schedule'][:12], 'frac', round(r['frac'],4), 'launches', r.get('launches'), 'busy', round(r.get('busy_ms') or 0,1))"
}
{
run c5 2 X=default
run c5 2 SGP_HYBRID_W=4096 SGP_HYBRID_WP=2048
run c5 2 SGP_HYBRID_W=8192 SGP_HYBRID_WP=2048
run target 2 SGP_HYBRID_W=4096 SGP_HYBRID_WP=2048
run n32k 3 SGP_HYBRID_W=4096 SGP_HYBRID_WP=2048
} 2>&1 | tee $OUT/hybrid_wp.txt
cd $R
SGP_HYBRID_WP=512 timeout 300 python -m pytest tests/test_gpu_dataflow.py -m gpu -q -x -p no:cacheprovider -k "hybrid_schedule_is_bit" > $OUT/pytest_wp.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_wp.log
