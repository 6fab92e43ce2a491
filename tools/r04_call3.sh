#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04c
Q="SGP_DATAFLOW=1,SGP_DF_FAT_MAX_N=0"
for N in 32768 65536; do
timeout 900 python tools/gpu_df_variants.py $N \
  "colmajor:$Q,SGP_DF_ORDER=0" \
  "colmajor tailprio:$Q,SGP_DF_ORDER=0,SGP_DF_TAILPRIO=1" \
  "q64x1 gang50:$Q,SGP_DF_ORDER=1,SGP_DF_PR=64,SGP_DF_PC=1,SGP_DF_GANG_US=50" \
  "q64x1 gang300:$Q,SGP_DF_ORDER=1,SGP_DF_PR=64,SGP_DF_PC=1,SGP_DF_GANG_US=300" \
  "q32x2 gang300:$Q,SGP_DF_ORDER=1,SGP_DF_PR=32,SGP_DF_PC=2,SGP_DF_GANG_US=300" \
  "q16x4 gang300:$Q,SGP_DF_ORDER=1,SGP_DF_PR=16,SGP_DF_PC=4,SGP_DF_GANG_US=300" \
  "q16x4 gang2000:$Q,SGP_DF_ORDER=1,SGP_DF_PR=16,SGP_DF_PC=4,SGP_DF_GANG_US=2000" \
  "q8x8 gang2000:$Q,SGP_DF_ORDER=1,SGP_DF_PR=8,SGP_DF_PC=8,SGP_DF_GANG_US=2000" \
  "q32x2 gang300 tailprio:$Q,SGP_DF_ORDER=1,SGP_DF_PR=32,SGP_DF_PC=2,SGP_DF_GANG_US=300,SGP_DF_TAILPRIO=1" \
  2>&1 | tee -a gpurun_out/r04c/variants.txt
done
# does the gang start produce L2 hits?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in "q16x4g SGP_DF_ORDER=1 SGP_DF_PR=16 SGP_DF_PC=4 SGP_DF_GANG_US=2000" "q64x1g SGP_DF_ORDER=1 SGP_DF_PR=64 SGP_DF_PC=1 SGP_DF_GANG_US=300"; do
  set -- $tag; t=$1; shift
  env "$@" timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/r04c/pmc_$t -o p -- \
      python $R/bench.py --config c3 --steps 1 --warmup 0 --cpu-sample 0 --no-host-api > /dev/null 2> $R/gpurun_out/r04c/pmc_$t.err
  f=$(find $R/gpurun_out/r04c/pmc_$t -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee $R/gpurun_out/r04c/pmc_$t.json
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
  rm -rf $R/gpurun_out/r04c/pmc_$t
  env "$@" SGP_DF_STATS=1 timeout 200 python $R/bench.py --config c3 --steps 2 --warmup 1 --cpu-sample 0 --no-host-api 2>&1 | grep -A2 "^dataflow order" | tail -3 | tee $R/gpurun_out/r04c/stats_$t.txt
done
