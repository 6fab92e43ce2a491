#!/bin/bash
# Round 5, second GPU call: (A) outer panel width of the gradient's bordered factorisation, (B) panel width / update groups of
# the sharded factorisation of the structured north-star model (chain-bound once the ranks are balanced).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05b
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for n in 16384 32768; do
  for w in default 1024 2048 4096; do
    if [ $w = default ]; then env -u SGP_WOUT timeout 200 python $R/tools/gpu_grad_split.py $n 2>&1 | grep N= | sed "s/^/WOUT=$w /"
    else SGP_WOUT=$w timeout 200 python $R/tools/gpu_grad_split.py $n 2>&1 | grep N= | sed "s/^/WOUT=$w /"; fi
  done
  SGP_LOOKAHEAD=0 timeout 200 python $R/tools/gpu_grad_split.py $n 2>&1 | grep N= | sed "s/^/LOOKAHEAD=0 /"
done > $OUT/grad_wout.txt 2>&1
cat $OUT/grad_wout.txt
for cfg in "1024 1 512" "1024 1 256" "512 1 256" "512 2 256" "512 1 0" "768 1 256" "768 1 384"; do
  set -- $cfg; W=$1; G=$2; S=$3
  tag=W${W}_G${G}_S${S}
  SGP_MULTI_PANEL=$W SGP_MULTI_GROUP=$G SGP_MULTI_SUBPANEL=$S timeout 300 python $R/tools/gpu_multi_profile.py target 8 $OUT/prof_target_$tag.json > $OUT/prof_target_$tag.log 2>&1
  python $R/tools/multi_projection.py $OUT/prof_target_$tag.json > $OUT/proj_target_$tag.txt 2>&1
  echo "== $tag"; grep -h "ownership\|allgather link 77 GB/s contend 1.00\|infinite\|serialised kernel\|panel factorisations" $OUT/proj_target_$tag.txt | cut -c1-260
  tail -1 $OUT/prof_target_$tag.log | cut -c1-200
done
