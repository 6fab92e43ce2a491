#!/bin/bash
mkdir -p gpurun_out
{
SGP_DF_TIMEOUT_S=3 timeout 300 python -m pytest tests/test_gpu_dataflow.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
run() { # config, extra env...
  c=$1; shift
  echo "== $c $*"
  env "$@" SGP_DF_STATS=1 SGP_DF_TIMEOUT_S=3 SGP_DATAFLOW=1 timeout 120 python bench.py --config $c --steps 2 --warmup 1 --cpu-sample 0 --no-host-api 2>&1 | grep -A1 "^dataflow" | tail -2
}
run c1 X=1
run c1 SGP_DF_FAT_MAX_N=0
run n4k X=1
run n4k SGP_DF_LOOKAHEAD=0
run c2 SGP_DF_LOOKAHEAD=0
run c2 SGP_DF_LOOKAHEAD=8
run c2 SGP_DF_LOOKAHEAD=16
run c2 SGP_DF_LOOKAHEAD=32
for c in c1 n4k c2 c3; do
  echo "== $c timing (no stats)"
  SGP_DF_TIMEOUT_S=3 SGP_DATAFLOW=1 timeout 200 python bench.py --config $c --steps 6 --warmup 2 --cpu-sample 0 --no-host-api 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('parity_rel'))"
done
} > gpurun_out/df3.txt 2>&1
cat gpurun_out/df3.txt
