#!/bin/bash
mkdir -p gpurun_out
{
timeout 20 tools/df_probe.bin 2048 128 512 2>&1 | grep -v watchdog | tail -3
SGP_DF_TIMEOUT_S=3 timeout 300 python -m pytest tests/test_gpu_dataflow.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5
run() { # config, extra env...
  c=$1; shift
  echo "== $c $*"
  env "$@" SGP_DF_STATS=1 SGP_DF_TIMEOUT_S=3 SGP_DATAFLOW=1 timeout 120 python bench.py --config $c --steps 2 --warmup 1 --cpu-sample 0 --no-host-api 2>&1 | grep -A1 "^dataflow" | tail -2
}
run c2 SGP_DF_LOOKAHEAD=0 SGP_DF_PARK=0
run c2 SGP_DF_LOOKAHEAD=0
run c2 SGP_DF_LOOKAHEAD=4
run c2 SGP_DF_LOOKAHEAD=16
run c3 SGP_DF_LOOKAHEAD=0
run c3 SGP_DF_LOOKAHEAD=16
} > gpurun_out/df4.txt 2>&1
cat gpurun_out/df4.txt
