#!/bin/bash
# Round 5, third GPU call: the near window of the sharded factorisation (G = 1: the panel after next in a launch of its own).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05c
mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_struct_zeros.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_multi.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_multi.log
cd /tmp; export TMPDIR=/tmp
for cfg in "target 1 512" "target 0 512" "target 1 256" "c5 1 512" "c5 0 512"; do
  set -- $cfg; C=$1; NW=$2; S=$3
  tag=${C}_NW${NW}_S${S}
  SGP_MULTI_NEAR=$NW SGP_MULTI_SUBPANEL=$S timeout 300 python $R/tools/gpu_multi_profile.py $C 8 $OUT/prof_$tag.json > $OUT/prof_$tag.log 2>&1
  python $R/tools/multi_projection.py $OUT/prof_$tag.json > $OUT/proj_$tag.txt 2>&1
  echo "== $tag"; grep -h "ownership\|allgather link 77 GB/s contend 1\|infinite\|serialised kernel\|panel factorisations" $OUT/proj_$tag.txt | cut -c1-300
  tail -1 $OUT/prof_$tag.log | cut -c1-220
done
for C in target c5; do
  timeout 300 python $R/bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config $C --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_${C}_multi8.json 2> $OUT/bench_${C}_multi8.err
  python -c "import json;d=json.load(open('$OUT/bench_${C}_multi8.json'));print('$C loopback-8', d['ms_per_step'], d['parity_rel'])"
done
