#!/bin/bash
# HBM-side traffic of the dominant kernel over a whole bench step (separate --pmc passes, MI355X_MICROARCH.md):
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on `bench.py --config <c> --steps 1`, averaged over the launches of
# sgp::gemm_nt_dma_kernel<1> / gemm_nt_dma_potrf_kernel<1, true> (the trailing updates).  -> gpurun_out/r02traffic/<c>_{FETCH_SIZE,WRITE_SIZE}.json
# usage: collect_traffic.sh [configs...]   (default: c5 target)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CONFIGS=${@:-c5 target}
for c in $CONFIGS; do
  for cnt in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $cnt --output-format csv -d $OUT/pmc_${c}_$cnt -o p -- \
        python $R/bench.py --config $c --steps 1 --warmup 0 --cpu-sample 0 --no-host-api > $OUT/${c}_$cnt.bench.json 2> $OUT/${c}_$cnt.err
    f=$(find $OUT/pmc_${c}_$cnt -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" $cnt > $OUT/${c}_$cnt.json <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ("gemm_nt_dma_kernel<1>" in r["Kernel_Name"] or "gemm_nt_dma_potrf_kernel<1," in r["Kernel_Name"]) and r["Counter_Name"] == sys.argv[2]]
v = [float(r["Counter_Value"]) for r in rows]
print(json.dumps({"counter": sys.argv[2], "launches": len(v), "sum_KiB": sum(v), "avg_KiB_per_launch": sum(v) / max(1, len(v))}))
PY
    cat $OUT/${c}_$cnt.json
    rm -rf $OUT/pmc_${c}_$cnt
  done
done
