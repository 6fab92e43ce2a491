#!/bin/bash
# round 6, call 3: uneven sub-panel pieces of the sharded sweep (+ the tests they touch)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call3
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multi_faults.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_multi.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_multi.log
grep -E "passed|failed|rc=" $OUT/pytest_multi.log | tail -3
prof() {   # tag config env...
  tag=$1; cfg=$2; shift 2
  env "$@" timeout 300 python $R/tools/gpu_multi_profile.py $cfg 8 $OUT/mp_${cfg}_$tag.json > $OUT/mp_${cfg}_$tag.log 2>&1
  python $R/tools/multi_projection.py $OUT/mp_${cfg}_$tag.json > $OUT/proj_${cfg}_$tag.txt 2>&1
  echo "== $cfg $tag"; tail -1 $OUT/mp_${cfg}_$tag.log; grep -E "allgather link 77 GB/s contend 1.00|infinite" $OUT/proj_${cfg}_$tag.txt
}
for cfg in target c5; do
  prof default $cfg X=1
  prof p768_256 $cfg SGP_MULTI_PIECES=768,256
  prof p512_256_256 $cfg SGP_MULTI_PIECES=512,256,256
  prof p512_384_128 $cfg SGP_MULTI_PIECES=512,384,128
  prof p640_256_128 $cfg SGP_MULTI_PIECES=640,256,128
  prof p896_128 $cfg SGP_MULTI_PIECES=896,128
done
prof w2048_p1024_512_256_256 target SGP_MULTI_PANEL=2048 SGP_MULTI_PIECES=1024,512,256,256
prof w1536_p768_512_256 target SGP_MULTI_PANEL=1536 SGP_MULTI_PIECES=768,512,256
ls $OUT | wc -l
