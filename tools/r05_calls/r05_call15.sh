#!/bin/bash
# Round 5: the hybrid schedule as a default -- tests, crossover sweep, the structured north-star lines under it
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05o
mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_dataflow.py tests/test_gpu_threads.py tests/test_gpu_struct_zeros.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -5 $OUT/pytest_a.log
cd /tmp
timeout 600 python $R/tools/gpu_hybrid_crossover.py 2>&1 | tee $OUT/crossover.txt
run() {
  c=$1; st=$2; shift 2
  env "$@" timeout 400 python $R/bench.py --config $c --steps $st --warmup 2 --cpu-sample 0 --no-host-api --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$c', '$*', round(d['ms_per_step'],3), 'ms', repr(d['logpdf']), d['parity_rel'], r['schedule'][:12], 'frac', round(r['frac'],4), 'launches', r.get('launches'), 'busy', r.get('busy_ms'))"
}
{
run target 2 SGP_HYBRID=0
run target 2 X=default
run target 2 SGP_HYBRID_WGS=512
run c3 5 SGP_HYBRID=0
run c3 5 X=default
run c5 2 X=default
run n32k 5 X=default
} 2>&1 | tee $OUT/lines.txt
