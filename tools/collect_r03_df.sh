#!/bin/bash
# Evidence for the dataflow factorisation (chol_df.hip) -> gpurun_out/r03df/ (one gpurun call).
# --kernel-trace/--stats and --pmc are separate runs.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03df
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in n4k c2; do timeout 300 python $R/bench.py --config $c --steps 30 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
for c in c3 n32k c4; do timeout 500 python $R/bench.py --config $c --steps 3 --warmup 1 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
# the launch-based schedules on the same box
for c in n4k c2 c3; do
  SGP_DATAFLOW=0 timeout 300 python $R/bench.py --config $c --steps 5 --warmup 2 --cpu-sample 0 --no-host-api > $OUT/bench_${c}_launches.json 2> $OUT/bench_${c}_launches.err
done
# per-workgroup time split + the per-column chain
for c in n4k c2 c3; do
  SGP_DF_STATS=1 timeout 200 python $R/bench.py --config $c --steps 2 --warmup 1 --cpu-sample 0 --no-host-api 2>&1 | grep -A1 "^dataflow" | tail -2
done > $OUT/df_stats.txt
for c in n4k c2 c3; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o $c -- \
      python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 --no-host-api > $OUT/prof_${c}_bench.json 2> $OUT/prof_$c.err
done
# HBM-side traffic + MFMA pipe of the one kernel, N = 16384 and 32768
for c in c2 c3; do
  for cnt in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $cnt | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $cnt --output-format csv -d $OUT/pmc_${c}_$tag -o p -- \
        python $R/bench.py --config $c --steps 1 --warmup 0 --cpu-sample 0 --no-host-api > $OUT/pmc_${c}_$tag.bench.json 2> $OUT/pmc_${c}_$tag.err
    f=$(find $OUT/pmc_${c}_$tag -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" > $OUT/pmc_${c}_$tag.json <<'PY'
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
    cat $OUT/pmc_${c}_$tag.json
    rm -rf $OUT/pmc_${c}_$tag
  done
done
timeout 400 python $R/tools/gpu_df_sizes.py 1024 2048 3072 4096 6144 8192 12288 16384 20480 24576 32768 49152 > $OUT/df_sizes.txt 2>&1
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
cat $OUT/df_stats.txt; tail -14 $OUT/df_sizes.txt
