#!/bin/bash
# Round 5: hybrid schedule, second sweep (wider panels; N = 65536; sizes in between)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05n
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() {
  c=$1; st=$2; shift 2
  env "$@" timeout 400 python $R/bench.py --config $c --steps $st --warmup 2 --cpu-sample 0 --no-host-api --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', '$*', round(d['ms_per_step'],3), 'ms', repr(d['logpdf']), d['parity_rel'], d['roofline']['schedule'][:12])"
}
{
run n32k 5 X=default
for W in 2048 4096; do for WG in 256 512; do
  run n32k 5 SGP_HYBRID=1 SGP_HYBRID_W=$W SGP_HYBRID_WGS=$WG SGP_HYBRID_FAT=1
done; done
run n32k 5 SGP_HYBRID=1 SGP_HYBRID_W=3072 SGP_HYBRID_WGS=256 SGP_HYBRID_FAT=1
run c2 20 X=default
for W in 2048 4096; do for WG in 256 512; do
  run c2 20 SGP_HYBRID=1 SGP_HYBRID_W=$W SGP_HYBRID_WGS=$WG SGP_HYBRID_FAT=1
done; done
run c5 2 X=default
for W in 2048 4096; do for WG in 256 512; do
  run c5 2 SGP_HYBRID=1 SGP_HYBRID_W=$W SGP_HYBRID_WGS=$WG SGP_HYBRID_FAT=1
done; done
} 2>&1 | tee $OUT/hybrid_sweep2.txt
cd $R
SGP_HYBRID=1 SGP_HYBRID_W=2048 SGP_HYBRID_WGS=256 SGP_HYBRID_FAT=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_golden.py tests/test_gpu_dataflow.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_hybrid.log 2>&1; echo "pytest under SGP_HYBRID rc=$?"; tail -3 $OUT/pytest_hybrid.log
