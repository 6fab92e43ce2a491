#!/bin/bash
# round 6, call 2: batch + fault tests, small-N aggregate timing, panel-width / instantiation variants of the sharded sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call2
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_multi_faults.py tests/test_bench_cli.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_new.log
tail -15 $OUT/pytest_new.log
timeout 600 python tools/gpu_batch_time.py 2048 4096 8192 > $OUT/batch_time.json 2> $OUT/batch_time.err; tail -3 $OUT/batch_time.err
SGP_BATCH_FAT=0 timeout 600 python tools/gpu_batch_time.py 2048 4096 8192 > $OUT/batch_time_lean.json 2> $OUT/batch_time_lean.err; tail -3 $OUT/batch_time_lean.err
prof() {   # tag config env...
  tag=$1; cfg=$2; shift 2
  env "$@" timeout 300 python $R/tools/gpu_multi_profile.py $cfg 8 $OUT/mp_${cfg}_$tag.json > $OUT/mp_${cfg}_$tag.log 2>&1
  python $R/tools/multi_projection.py $OUT/mp_${cfg}_$tag.json > $OUT/proj_${cfg}_$tag.txt 2>&1
  echo "== $cfg $tag"; tail -1 $OUT/mp_${cfg}_$tag.log; grep -E "allgather link 77 GB/s contend 1.00|infinite" $OUT/proj_${cfg}_$tag.txt
}
for cfg in target c5; do
  prof default $cfg X=1
  prof lean512 $cfg SGP_HYBRID_FAT=0 SGP_HYBRID_WGS=512
  prof w512_s256 $cfg SGP_MULTI_PANEL=512 SGP_MULTI_SUBPANEL=256
  prof w512_s0 $cfg SGP_MULTI_PANEL=512 SGP_MULTI_SUBPANEL=0
  prof w512_s256_g2 $cfg SGP_MULTI_PANEL=512 SGP_MULTI_SUBPANEL=256 SGP_MULTI_GROUP=2
  prof w512_s0_g2 $cfg SGP_MULTI_PANEL=512 SGP_MULTI_SUBPANEL=0 SGP_MULTI_GROUP=2
  prof w768_s384 $cfg SGP_MULTI_PANEL=768 SGP_MULTI_SUBPANEL=384
done
ls $OUT | wc -l
