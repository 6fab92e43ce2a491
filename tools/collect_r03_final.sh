#!/bin/bash
# Final round-3 refresh of the evidence that the round's last code changes touch (one gpurun call) -> gpurun_out/r03final,
# on top of what tools/collect_r03.sh collected earlier in the round (isolated-GEMM / assembly PMC passes, multi-GPU profiles,
# prediction / gradient / fp32 timings: code unchanged since).  --kernel-trace/--stats and --pmc are separate runs.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03final
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|rc=" $OUT/pytest_gpu.log | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
for c in c1 n4k c2; do timeout 300 python $R/bench.py --config $c --steps 30 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
for c in c3 n32k c4 c5 target; do timeout 500 python $R/bench.py --config $c --steps 3 --warmup 1 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
for c in n4k c2 c3; do
  SGP_DATAFLOW=0 timeout 300 python $R/bench.py --config $c --steps 5 --warmup 2 --cpu-sample 0 --no-host-api > $OUT/bench_${c}_launches.json 2> $OUT/bench_${c}_launches.err
done
for c in c5 target; do
  SGP_LOOKAHEAD=2 timeout 300 python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 --no-host-api > $OUT/bench_${c}_lookahead.json 2> $OUT/bench_${c}_lookahead.err
done
timeout 300 python $R/bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config c5 --steps 2 --warmup 1 > $OUT/bench_c5_multi8_loopback.json 2> $OUT/bench_c5_multi8_loopback.err
timeout 300 python $R/bench.py --gpus 2 --devices 0,0 --config c4 --steps 2 --warmup 1 > $OUT/bench_c4_multi2_loopback.json 2> $OUT/bench_c4_multi2_loopback.err
for c in n4k c2 c3; do
  SGP_DF_STATS=1 timeout 200 python $R/bench.py --config $c --steps 2 --warmup 1 --cpu-sample 0 --no-host-api 2>&1 | grep -A1 "^dataflow" | tail -2
done > $OUT/df_stats.txt
for c in c5 target c3 c2 n4k c1; do
  st=3; [ $c = c1 ] && st=10
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o $c -- \
      python $R/bench.py --config $c --steps $st --warmup 1 --cpu-sample 0 --no-host-api > $OUT/prof_${c}_bench.json 2> $OUT/prof_$c.err
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -o c4 -- \
    python $R/bench.py --config c4 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/prof_c4_bench.json 2> $OUT/prof_c4.err
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
head -c 300 $OUT/bench_c5.json; echo; cat $OUT/df_stats.txt | cut -c1-200
