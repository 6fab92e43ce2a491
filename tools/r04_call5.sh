#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04e; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_bench_cli.py tests/test_gpu_dist.py -x -q 2>&1 | tail -25 ) > $O/pytest_multi.txt
tail -4 $O/pytest_multi.txt
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
for v in "w1024_g1 SGP_MULTI_GROUP=1 SGP_MULTI_PANEL_TAIL=0" "w1024_g2 SGP_MULTI_GROUP=2 SGP_MULTI_PANEL_TAIL=0" "w1024_g4 SGP_MULTI_GROUP=4 SGP_MULTI_PANEL_TAIL=0" \
         "w512_g4 SGP_MULTI_PANEL=512 SGP_MULTI_GROUP=4 SGP_MULTI_PANEL_TAIL=0" "w512_g8 SGP_MULTI_PANEL=512 SGP_MULTI_GROUP=8 SGP_MULTI_PANEL_TAIL=0" \
         "mixed_g4 SGP_MULTI_GROUP=4" "mixed_g2 SGP_MULTI_GROUP=2" "w1024t512f50_g4 SGP_MULTI_TAIL_FRAC=0.5" ; do
  set -- $v; tag=$1; shift
  run $tag "$@"
  prof $tag "$@"
done
