#!/bin/bash
# Round-3 evidence for profiles/ (one gpurun call).  --kernel-trace/--stats and --pmc are separate runs.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03final
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|rc=" $OUT/pytest_gpu.log | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
for c in c1 n4k c2; do timeout 300 python $R/bench.py --config $c --steps 30 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
for c in c3 c4 c5 target; do timeout 500 python $R/bench.py --config $c --steps 3 --warmup 1 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
# both schedules at N = 65536 (serial is the default there), same box
for c in c5 target; do
  SGP_LOOKAHEAD=2 timeout 300 python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 --no-host-api > $OUT/bench_${c}_lookahead.json 2> $OUT/bench_${c}_lookahead.err
done
# the in-process multi-GPU context through bench.py's own --gpus path (loopback ranks on the one GPU)
timeout 300 python $R/bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config c5 --steps 2 --warmup 1 > $OUT/bench_c5_multi8_loopback.json 2> $OUT/bench_c5_multi8_loopback.err
timeout 300 python $R/bench.py --gpus 2 --devices 0,0 --config c4 --steps 2 --warmup 1 > $OUT/bench_c4_multi2_loopback.json 2> $OUT/bench_c4_multi2_loopback.err
timeout 300 python $R/bench.py --config c5 --force-dist --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_c5_dist1.json 2> $OUT/bench_c5_dist1.err
for c in c5 c2 c1 target; do
  st=3; [ $c = c1 ] && st=10
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o $c -- \
      python $R/bench.py --config $c --steps $st --warmup 1 --cpu-sample 0 --no-host-api > $OUT/prof_${c}_bench.json 2> $OUT/prof_$c.err
  f=$(find $OUT/prof_$c -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && timeout 60 python $R/tools/timeline_busy.py $f > $OUT/timeline_$c.txt 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -o c4 -- \
    python $R/bench.py --config c4 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/prof_c4_bench.json 2> $OUT/prof_c4.err

for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_gemm1_$c -o g -- python $R/tools/gpu_gemm_one.py > $OUT/pmc_gemm1_$c.log 2>&1
done
timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_gemm1_SQ -o g -- python $R/tools/gpu_gemm_one.py > $OUT/pmc_gemm1_SQ.log 2>&1
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_gemm1_TCC -o g -- python $R/tools/gpu_gemm_one.py > $OUT/pmc_gemm1_TCC.log 2>&1
for c in c2 c3 c5 target; do timeout 200 python $R/tools/gpu_assemble_one.py $c 10 2>&1 | tail -1; done > $OUT/assemble_times.txt
for cset in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $cset | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $cset --output-format csv -d $OUT/asm_pmc_$tag -o a -- python $R/tools/gpu_assemble_one.py c5 3 > $OUT/asm_pmc_$tag.log 2>&1
done
timeout 120 python $R/tools/gpu_gemm_sizes.py > $OUT/gemm_sizes.log 2>&1
timeout 200 python $R/tools/gpu_grad_time.py 16384 > $OUT/grad_time.log 2>&1
timeout 120 python $R/tools/gpu_predict_time.py 16384x16384 > $OUT/predict_time.log 2>&1
timeout 200 python $R/tools/gpu_illcond.py > $OUT/illcond.log 2>&1
timeout 400 python $R/tools/gpu_multi_time.py c2 c5 > $OUT/multi_time.log 2>&1
timeout 400 python $R/tools/gpu_multi_ops_time.py 16384 4 > $OUT/multi_ops_time.log 2>&1
for P in 8 4 2; do timeout 600 python $R/tools/gpu_multi_profile.py c5 $P $OUT/multi_profile_c5_P$P.json 2>&1 | tail -1; done > $OUT/multi_profile.log
timeout 300 python $R/tools/gpu_multi_profile.py target 8 $OUT/multi_profile_target_P8.json 2>&1 | tail -1 >> $OUT/multi_profile.log
timeout 300 python $R/tools/gpu_f32_time.py 2048 4096 16384 32768 65536 > $OUT/f32_time.log 2>&1
timeout 300 python $R/bench.py --config c5 --dtype f32 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_c5_f32.json 2> $OUT/bench_c5_f32.err
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
head -c 300 $OUT/bench_c5.json; echo; tail -3 $OUT/multi_ops_time.log
