#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04f; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_bench_cli.py tests/test_gpu_dist.py -q 2>&1 | tail -25 ) > $O/pytest_multi.txt
grep -E "passed|failed" $O/pytest_multi.txt | tail -3
D8=0,0,0,0,0,0,0,0
run() { tag=$1; shift
  env "$@" timeout 600 python bench.py --gpus 8 --devices $D8 --config c5 --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_c5_lb8_$tag.json 2> $O/bench_c5_lb8_$tag.err
  python -c "
import json; d=json.load(open('$O/bench_c5_lb8_$tag.json')); print('$tag', 'ms_per_step', round(d['ms_per_step'],1), 'host enqueue', round(d['multi_gpu']['host_enqueue_ms'],1), 'parity', d['parity_rel'])" || tail -3 $O/bench_c5_lb8_$tag.err
}
prof() { tag=$1; shift
  env "$@" timeout 600 python tools/gpu_multi_profile.py c5 8 $O/multi_profile_c5_P8_$tag.json 2>&1 | tail -1
  python tools/multi_projection.py $O/multi_profile_c5_P8_$tag.json | tee $O/projection_c5_P8_$tag.txt | grep -v direct
}
for v in "w1024_sub0 SGP_MULTI_SUBPANEL=0" "w1024_sub256 SGP_MULTI_SUBPANEL=256" "w1024_sub512 SGP_MULTI_SUBPANEL=512" "w1024_sub128 SGP_MULTI_SUBPANEL=128" \
         "w2048_sub256 SGP_MULTI_PANEL=2048 SGP_MULTI_SUBPANEL=256" "w2048_sub512 SGP_MULTI_PANEL=2048 SGP_MULTI_SUBPANEL=512" \
         "w1024_sub256_g4 SGP_MULTI_GROUP=4" ; do
  set -- $v; tag=$1; shift
  run $tag SGP_MULTI_PANEL_TAIL=0 "$@"
  prof $tag SGP_MULTI_PANEL_TAIL=0 "$@"
done
