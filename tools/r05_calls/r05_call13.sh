#!/bin/bash
# Round 5: the hybrid schedule (round-4 verdict, item 4) -- dataflow PANELS + lock-step trailing updates with look-ahead.
# Bit-identity against the default schedule, then a sweep of panel width / workgroups / fat-lean at c2 and n32k.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05m
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
run() {  # config steps env...
  c=$1; st=$2; shift 2
  env "$@" timeout 300 python $R/bench.py --config $c --steps $st --warmup 3 --cpu-sample 0 --no-host-api --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', '$*', round(d['ms_per_step'],3), 'ms', repr(d['logpdf']), d['parity_rel'], d['roofline']['schedule'][:12])"
}
{
run c2 20 X=default
run c2 20 SGP_DATAFLOW=0
for W in 512 1024 2048; do for WG in 64 128 256; do for F in 0 1; do
  run c2 20 SGP_HYBRID=1 SGP_HYBRID_W=$W SGP_HYBRID_WGS=$WG SGP_HYBRID_FAT=$F
done; done; done
run n32k 5 X=default
run n32k 5 SGP_DATAFLOW=0
for W in 1024 2048; do for WG in 128 256 512; do for F in 0 1; do
  run n32k 5 SGP_HYBRID=1 SGP_HYBRID_W=$W SGP_HYBRID_WGS=$WG SGP_HYBRID_FAT=$F
done; done; done
run n4k 30 X=default
run n4k 30 SGP_HYBRID=1 SGP_HYBRID_W=1024 SGP_HYBRID_WGS=128 SGP_HYBRID_FAT=1
} 2>&1 | tee $OUT/hybrid_sweep.txt
