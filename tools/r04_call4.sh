#!/bin/bash
# round 4, call 4: the grouped / batched multi-GPU schedule -- correctness, loopback-8 timing, serialised profiles
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04d; mkdir -p $O
( timeout 1500 python -m pytest tests/test_gpu_multi.py tests/test_bench_cli.py tests/test_gpu_dist.py -x -q 2>&1 | tail -25 ) > $O/pytest_multi.txt
tail -6 $O/pytest_multi.txt
D8=0,0,0,0,0,0,0,0
run() { tag=$1; shift
  env "$@" timeout 600 python bench.py --gpus 8 --devices $D8 --config c5 --steps 2 --warmup 1 --cpu-sample 0 > $O/bench_c5_lb8_$tag.json 2> $O/bench_c5_lb8_$tag.err
  python -c "
import json; d=json.load(open('$O/bench_c5_lb8_$tag.json')); print('$tag', 'ms_per_step', round(d['ms_per_step'],1), 'parity', d['parity_rel'], 'per-rank TF/s', [round(x,1) if x else None for x in d['roofline']['per_rank_update_tflops']])" || tail -3 $O/bench_c5_lb8_$tag.err
}
run default
run g1_uniform SGP_MULTI_GROUP=1 SGP_MULTI_PANEL_TAIL=0
run g4_uniform SGP_MULTI_PANEL_TAIL=0
run g2_mixed SGP_MULTI_GROUP=2
timeout 300 python bench.py --config c5 --steps 2 --warmup 1 --cpu-sample 0 --no-host-api > $O/bench_c5_single.json 2> $O/bench_c5_single.err
python -c "
import json; d=json.load(open('$O/bench_c5_single.json')); print('single-GPU c5 ms_per_step', round(d['ms_per_step'],1))"
prof() { tag=$1; shift
  env "$@" timeout 600 python tools/gpu_multi_profile.py c5 8 $O/multi_profile_c5_P8_$tag.json 2>&1 | tail -1
  python tools/multi_projection.py $O/multi_profile_c5_P8_$tag.json | tee $O/projection_c5_P8_$tag.txt
}
prof default
prof g1_uniform SGP_MULTI_GROUP=1 SGP_MULTI_PANEL_TAIL=0
prof g4_uniform SGP_MULTI_PANEL_TAIL=0
