#!/bin/bash
# Round-2 GPU pass 1 (through gpurun): full -m gpu suite, every bench config, rocprofv3 kernel stats.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for c in c1 n4k c2; do timeout 300 python $R/bench.py --config $c --steps 20 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
for c in c3 c4 c5 target; do timeout 500 python $R/bench.py --config $c --steps 3 --warmup 1 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
timeout 300 python $R/bench.py --config c2 --force-dist --steps 5 --warmup 1 --cpu-sample 0 > $OUT/bench_c2_nccl1.json 2> $OUT/bench_c2_nccl1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o c5 -- \
    python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-host-api > $OUT/prof_c5_bench.json 2> $OUT/prof_c5.err
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c2 -o c2 -- \
    python $R/bench.py --config c2 --steps 3 --warmup 1 --cpu-sample 0 --no-host-api > $OUT/prof_c2_bench.json 2> $OUT/prof_c2.err
for c in c1 n4k; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o $c -- \
      python $R/bench.py --config $c --steps 5 --warmup 2 --cpu-sample 0 --no-host-api > $OUT/prof_${c}_bench.json 2> $OUT/prof_$c.err
  f=$(find $OUT/prof_$c -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && timeout 60 python $R/tools/timeline_busy.py $f > $OUT/timeline_$c.txt 2>&1
  [ -n "$f" ] && tail -200 $f > $OUT/trace_tail_$c.csv
done
# knob experiments (one summary line each)
for v in SGP_PANEL_PRIO=3 SGP_PANEL_PRIO=1 SGP_LOOKAHEAD=0 SGP_WOUT=256 SGP_WOUT=1024; do
  for c in c1 n4k c2; do
    echo -n "$v $c " >> $OUT/knobs.txt
    env $v timeout 200 python $R/bench.py --config $c --steps 20 --warmup 3 --cpu-sample 0 --no-host-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['parity_rel'])" >> $OUT/knobs.txt
  done
done
for v in SGP_PANEL_PRIO=3 SGP_PANEL_PRIO=1; do
  echo -n "$v c5 " >> $OUT/knobs.txt
  env $v timeout 300 python $R/bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-host-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['parity_rel'], d['roofline']['frac'])" >> $OUT/knobs.txt
done
cat $OUT/knobs.txt
for c in c5 c2; do f=$(find $OUT/prof_$c -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && timeout 60 python $R/tools/timeline_busy.py $f > $OUT/timeline_$c.txt 2>&1; done
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
for c in c1 n4k c2 c3 c4 c5 target c2_nccl1; do echo "== $c"; head -c 600 $OUT/bench_$c.json; echo; tail -2 $OUT/bench_$c.err; done
