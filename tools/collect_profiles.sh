#!/bin/bash
# Collect the rocprofv3 evidence + bench lines for profiles/ on the GPU box (run through gpurun).
# --kernel-trace/--stats and --pmc are separate runs (never combined with other trace domains).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py > $OUT/bench_c5.json 2> $OUT/bench_c5.err
for c in c1 c2 c3 c4; do timeout 300 python $R/bench.py --config $c > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o c5 -- \
    python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/prof_c5_bench.json 2> $OUT/prof_c5.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -o c2 -- \
    python $R/bench.py --config c2 --steps 3 --warmup 1 --cpu-sample 0 > $OUT/prof_c2_bench.json 2> $OUT/prof_c2.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_gemm1_$c -o g -- python $R/tools/gpu_gemm_one.py > $OUT/pmc_gemm1_$c.log 2>&1
done
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_gemm1_SQ -o g -- python $R/tools/gpu_gemm_one.py > $OUT/pmc_gemm1_SQ.log 2>&1
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_gemm1_TCC -o g -- python $R/tools/gpu_gemm_one.py > $OUT/pmc_gemm1_TCC.log 2>&1
timeout 120 python $R/tools/gpu_sustained.py 400000 2> $OUT/mfma_variants.log
timeout 120 python $R/tools/gpu_gemm_abl.py > $OUT/gemm_variants.log 2>&1
for c in c5 c2; do f=$(find $OUT/prof_$c -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && timeout 60 python $R/tools/timeline_busy.py $f > $OUT/timeline_$c.txt 2>&1; done
timeout 120 python $R/tools/gpu_gemm_sizes.py > $OUT/gemm_sizes.log 2>&1
timeout 200 python $R/tools/gpu_grad_time.py 16384 > $OUT/grad_time.log 2>&1
timeout 120 python $R/tools/gpu_predict_time.py 16384x16384 > $OUT/predict_time.log 2>&1
timeout 200 python $R/tools/gpu_illcond.py > $OUT/illcond.log 2>&1
rm -f $OUT/*/*kernel_trace.csv   # large; the stats CSV is what gets committed
grep -h gemm $OUT/pmc_gemm1_FETCH_SIZE.log | head -2
head -c 400 $OUT/bench_c5.json
