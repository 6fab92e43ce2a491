#!/bin/bash
# multi-GPU driver at scale on the one GPU: loopback ranks, c5 (N = 65536) -- overheads vs the single-GPU driver,
# serialised per-panel profile for the 8-GPU projection, bench line through `bench.py --gpus N --devices 0,0,..`
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03c
mkdir -p $OUT
cd $R
for P in 8 4 2; do
  timeout 600 python tools/gpu_multi_profile.py c5 $P $OUT/multi_profile_c5_P$P.json 2>&1 | tail -1
done
timeout 300 python tools/gpu_multi_profile.py target 8 $OUT/multi_profile_target_P8.json 2>&1 | tail -1
timeout 300 python tools/gpu_multi_profile.py c5 8 $OUT/multi_profile_c5_P8_w512.json 512 2>&1 | tail -1
timeout 300 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config c5 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_c5_loop8.json 2> $OUT/bench_c5_loop8.err; head -c 300 $OUT/bench_c5_loop8.json; echo
SGP_MULTI_BCAST=direct timeout 300 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config c5 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_c5_loop8_direct.json 2> $OUT/bench_c5_loop8_direct.err; head -c 300 $OUT/bench_c5_loop8_direct.json; echo
