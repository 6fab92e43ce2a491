#!/bin/bash
# round 4, call 2: WHY the XCD-affine queues lose -- per-workgroup time split, L2 hit rate, fabric traffic, MFMA pipe per order
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
c=${1:-c3}
run() { # tag, env...
  tag=$1; shift
  env "$@" SGP_DF_STATS=1 timeout 200 python $R/bench.py --config $c --steps 2 --warmup 1 --cpu-sample 0 --no-host-api 2>&1 | grep -A2 "^dataflow" | tail -3 > $OUT/stats_${c}_$tag.txt
  env "$@" timeout 200 python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 --no-host-api > $OUT/bench_${c}_$tag.json 2> $OUT/bench_${c}_$tag.err
  for cnt in FETCH_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
    t2=$(echo $cnt | cut -d' ' -f1)
    env "$@" timeout 300 rocprofv3 --pmc $cnt --output-format csv -d $OUT/pmc_${c}_${tag}_$t2 -o p -- \
        python $R/bench.py --config $c --steps 1 --warmup 0 --cpu-sample 0 --no-host-api > /dev/null 2> $OUT/pmc_${c}_${tag}_$t2.err
    f=$(find $OUT/pmc_${c}_${tag}_$t2 -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" > $OUT/pmc_${c}_${tag}_$t2.json <<'PY'
import csv, json, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "chol_dataflow" not in r["Kernel_Name"]:
        continue
    a = acc.setdefault(r["Counter_Name"], [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
print(json.dumps({k: {"launches": v[0], "avg_per_launch": v[1] / max(1, v[0])} for k, v in acc.items()}))
PY
    rm -rf $OUT/pmc_${c}_${tag}_$t2
  done
  echo "== $c $tag"; cat $OUT/stats_${c}_$tag.txt; python -c "
import json,sys
d=json.load(open('$OUT/bench_${c}_$tag.json')); print('ms_per_step', d['ms_per_step'])"; cat $OUT/pmc_${c}_${tag}_*.json
}
run colmajor SGP_DF_ORDER=0
run q64x1 SGP_DF_ORDER=1 SGP_DF_PR=64 SGP_DF_PC=1
run q32x2 SGP_DF_ORDER=1 SGP_DF_PR=32 SGP_DF_PC=2
run q8x8 SGP_DF_ORDER=1 SGP_DF_PR=8 SGP_DF_PC=8
