#!/bin/bash
# Round 5: structural zeros inside the panel solves / a packed panel's own factorisation -- suites, then the north-star model on
# one GPU and the 8-rank profile + projection (sub-panels of 512 and 256).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05j
mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests/test_gpu_struct_zeros.py tests/test_gpu_multi.py tests/test_gpu_fused_potrf.py tests/test_gpu_baseline_golden.py tests/test_gpu_random_programmes.py tests/test_gpu_dist.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp; export TMPDIR=/tmp
for c in target c3; do
  timeout 300 python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', round(d['ms_per_step'],2), 'ms; host API', round(d['host_api']['ms_per_call'],2), d['parity_rel'], d['executed_work_fraction'])"
done | tee $OUT/single.txt
for S in 512 256; do
  SGP_MULTI_SUBPANEL=$S timeout 400 python $R/tools/gpu_multi_profile.py target 8 $OUT/prof_target_S$S.json > $OUT/prof_target_S$S.log 2>&1
  python $R/tools/multi_projection.py $OUT/prof_target_S$S.json > $OUT/proj_target_S$S.txt 2>&1
  echo "== S=$S"; grep -h "ownership\|allgather link 77 GB/s contend 1\|infinite\|serialised kernel\|panel factorisations" $OUT/proj_target_S$S.txt | cut -c1-300
  tail -1 $OUT/prof_target_S$S.log | cut -c1-220
done
timeout 300 python $R/bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config target --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('target loopback-8', d['ms_per_step'], d['parity_rel'])"
