#!/bin/bash
# GPU pass 3: assembly kernel (two rows / thread, 16-byte stores, live tiles only): parity + timing + PMC
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_golden.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for c in c2 c3 c5 target; do timeout 200 python $R/tools/gpu_assemble_one.py $c 10 2>&1 | tail -1; done | tee $OUT/assemble_times.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/asm_stats -o a -- python $R/tools/gpu_assemble_one.py c5 10 > $OUT/asm_stats.log 2>&1
for cset in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $cset | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $cset --output-format csv -d $OUT/asm_pmc_$tag -o a -- python $R/tools/gpu_assemble_one.py c5 3 > $OUT/asm_pmc_$tag.log 2>&1
  f=$(find $OUT/asm_pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' | tee -a $OUT/asm_pmc_summary.txt
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(list)
for r in rows:
    if 'assemble_block' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k, 'per-launch avg', sum(v)/len(v), 'n', len(v))
PY
done
one() { echo -n "$1 $2 " ; env $1 timeout 300 python $R/bench.py --config $2 --steps $3 --warmup 2 --cpu-sample 0 --no-host-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['parity_rel'], round(d['roofline']['frac'],3), round(d['stages']['kernelmatrix_GBps']), round(d['stages']['assemble_ms'],3))"; }
for c in c1 n4k c2; do one X=1 $c 20; done | tee $OUT/bench_small.txt
for c in c3 c5; do one X=1 $c 3; done | tee -a $OUT/bench_small.txt
rm -rf $OUT/asm_stats/*/*kernel_trace.csv
