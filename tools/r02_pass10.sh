#!/bin/bash
# GPU pass 10 (one gpurun call, <= 13 min): the fused update + potrf_diag launches (SGP_FUSE_POTRF).
#   A  bench c1 / n4k / c2 with the knob at 0 / 3 / 11 (ms per logpdf + parity against the CPU goldens)
#   B  tests/test_gpu_fused_potrf.py (bit identity of every operator, PosDef info)
#   C  tools/pick_fuse.py -> the value the rest of the script (and, if it wins, the library default) uses
#   E  knob sweep around it (panel width, wave priority, panel_solve workgroup count, look-ahead)
#   D  the whole -m gpu suite with the chosen value exported
#   F  rocprofv3 kernel stats + stream occupancy of c2 / n4k / c1 with it
#   G  c3 / c5 / target lines
# Every part checks the clock: what does not fit is skipped, results are written as they come.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02fuse
mkdir -p $OUT
T0=$(date +%s)
LIMIT=${LIMIT:-760}
left() { echo $(( LIMIT - ( $(date +%s) - T0 ) )); }
stamp() { echo "== $1 at $(( $(date +%s) - T0 )) s" | tee -a $OUT/progress.txt; }
cd /tmp && export TMPDIR=/tmp
one() {  # env config steps
  echo -n "$1 $2 "
  env $1 timeout 120 python $R/bench.py --config $2 --steps $3 --warmup 3 --cpu-sample 0 --no-host-api 2>>$OUT/bench_err.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(round(d['ms_per_step'],4), d['parity_rel'], r.get('frac'))" 2>/dev/null || echo "FAILED"
}
stamp A
for c in c1 n4k c2; do for f in 0 3 11; do st=30; [ $c = c2 ] && st=15; one "SGP_FUSE_POTRF=$f" $c $st; done; done | tee $OUT/fuse.txt
stamp B
cd $R
timeout 150 python -m pytest tests/test_gpu_fused_potrf.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_fused.log 2>&1; echo "rc=$?" >> $OUT/pytest_fused.log
tail -4 $OUT/pytest_fused.log
cd /tmp
stamp C
FUSE=$(python $R/tools/pick_fuse.py $OUT/fuse.txt $OUT/pytest_fused.log | tee $OUT/choice.txt | tail -1)
echo "chosen SGP_FUSE_POTRF=$FUSE"
export SGP_FUSE_POTRF=$FUSE
stamp E
{
  for w in 256 1024; do one "SGP_WOUT=$w" c2 15; done
  one "SGP_PANEL_PRIO=3" c2 15
  for d in 128 64; do one "SGP_PS_DIV=$d" c2 15; done
  for w in 2048 4096; do one "SGP_WOUT=$w" n4k 30; done
  one "SGP_LOOKAHEAD=0" n4k 30
  one "SGP_PANEL_PRIO=3" n4k 30
  one "SGP_PANEL_PRIO=3" c1 30
} | tee $OUT/knobs.txt
stamp D
cd $R
if [ $(left) -gt 200 ]; then
  tl=$(( $(left) - 170 )); [ $tl -gt 330 ] && tl=330
  timeout $tl python -m pytest tests -m gpu -q -n 4 --dist loadfile -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? (SGP_FUSE_POTRF=$FUSE, limit $tl s)" >> $OUT/pytest_gpu.log
  tail -6 $OUT/pytest_gpu.log
fi
cd /tmp
stamp F
for c in c2 n4k c1; do
  [ $(left) -lt 100 ] && break
  st=3; [ $c = c1 ] && st=10; [ $c = n4k ] && st=10
  timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o $c -- \
      python $R/bench.py --config $c --steps $st --warmup 2 --cpu-sample 0 --no-host-api > $OUT/prof_${c}_bench.json 2> $OUT/prof_$c.err
  f=$(find $OUT/prof_$c -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && timeout 30 python $R/tools/timeline_busy.py $f > $OUT/timeline_$c.txt 2>&1
  s=$(find $OUT/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$s" ] && head -9 $s
done
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
stamp G
{
  [ $(left) -gt 60 ] && one "SGP_FUSE_POTRF=$FUSE" c3 5
  [ $(left) -gt 60 ] && one "SGP_FUSE_POTRF=$FUSE" c5 3
  [ $(left) -gt 60 ] && one "SGP_FUSE_POTRF=0" c5 3
  [ $(left) -gt 60 ] && one "SGP_FUSE_POTRF=$FUSE" target 3
  [ $(left) -gt 60 ] && one "SGP_FUSE_POTRF=0 SGP_LOOKAHEAD=0" c5 3
  [ $(left) -gt 40 ] && one "SGP_FUSE_POTRF=0" c3 5
} | tee $OUT/big.txt
# full bench lines (with host_api) of the small configs for profiles/
for c in c1 n4k c2; do
  [ $(left) -lt 30 ] && break
  timeout 90 python $R/bench.py --config $c --steps 30 --warmup 3 --cpu-sample 8192 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
done
stamp end
