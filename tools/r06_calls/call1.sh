#!/bin/bash
# round 6, call 1: GPU suite on the new sharded panel kernel + serialised profiles of the variants (one box: same-box A/B)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call1
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
prof() {   # tag config env...
  tag=$1; cfg=$2; shift 2
  env "$@" timeout 300 python $R/tools/gpu_multi_profile.py $cfg 8 $OUT/mp_${cfg}_$tag.json > $OUT/mp_${cfg}_$tag.log 2>&1
  python $R/tools/multi_projection.py $OUT/mp_${cfg}_$tag.json > $OUT/proj_${cfg}_$tag.txt 2>&1
  echo "== $cfg $tag"; tail -1 $OUT/mp_${cfg}_$tag.log; grep -E "allgather link 77 GB/s contend 1.00|infinite|panel factorisations" $OUT/proj_${cfg}_$tag.txt
}
for cfg in target c5; do
  prof r5chain $cfg SGP_MULTI_PANEL_DF=0
  prof df_nofuse $cfg SGP_MULTI_FUSE_LA=0
  prof default $cfg X=1
  prof sub256 $cfg SGP_MULTI_SUBPANEL=256
  prof sub0 $cfg SGP_MULTI_SUBPANEL=0
done
prof nocompact target SGP_MULTI_COMPACT=0
prof sub256_w2048 target SGP_MULTI_SUBPANEL=256 SGP_MULTI_PANEL=2048
prof sub128 target SGP_MULTI_SUBPANEL=128
ls $OUT | wc -l
