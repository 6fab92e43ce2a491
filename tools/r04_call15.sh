#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04j; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_bench_cli.py -q -x 2>&1 | tail -15 ) > $O/pytest_multi.txt
grep -E "passed|failed" $O/pytest_multi.txt | tail -2; grep -E "^FAILED|Error" $O/pytest_multi.txt | head -5
D8=0,0,0,0,0,0,0,0
for v in "t1_sub512 SGP_MULTI_THREADS=1" "t0_sub512 SGP_MULTI_THREADS=0" "t1_sub256 SGP_MULTI_THREADS=1 SGP_MULTI_SUBPANEL=256" "t0_sub256 SGP_MULTI_THREADS=0 SGP_MULTI_SUBPANEL=256" "t1_sub0 SGP_MULTI_THREADS=1 SGP_MULTI_SUBPANEL=0" "t0_sub0 SGP_MULTI_THREADS=0 SGP_MULTI_SUBPANEL=0"; do
  set -- $v; tag=$1; shift
  env "$@" timeout 600 python bench.py --gpus 8 --devices $D8 --config c5 --steps 3 --warmup 1 --cpu-sample 0 > $O/bench_c5_lb8_$tag.json 2> $O/bench_c5_lb8_$tag.err
  python -c "
import json; d=json.load(open('$O/bench_c5_lb8_$tag.json')); print('$tag', 'ms_per_step', round(d['ms_per_step'],1), 'host enqueue (slowest thread / the one thread)', round(d['multi_gpu']['host_enqueue_ms'],1), 'parity', d['parity_rel'])" || tail -3 $O/bench_c5_lb8_$tag.err
done
