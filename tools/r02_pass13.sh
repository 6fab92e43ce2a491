#!/bin/bash
# GPU pass 13 (final collection of the round, <= 7 min): the -m gpu suite with the final defaults, then the evidence for
# profiles/ in order of importance: the c5 line, its rocprofv3 kernel stats + stream occupancy, its FETCH / WRITE
# passes, the other bench lines.  Every part checks the clock.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02final2
mkdir -p $OUT
T0=$(date +%s)
LIMIT=${LIMIT:-430}
left() { echo $(( LIMIT - ( $(date +%s) - T0 ) )); }
stamp() { echo "== $1 at $(( $(date +%s) - T0 )) s" | tee -a $OUT/progress.txt; }
stamp suite
cd $R
timeout 200 python -m pytest tests -m gpu -q -n 4 --dist loadfile -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
line() { timeout 200 python $R/bench.py --config $1 --steps $2 --warmup $3 > $OUT/bench_$1.json 2> $OUT/bench_$1.err; head -c 330 $OUT/bench_$1.json; echo; }
stamp c5
line c5 3 1
stamp prof_c5
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o c5 -- \
    python $R/bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-host-api > $OUT/prof_c5_bench.json 2> $OUT/prof_c5.err
f=$(find $OUT/prof_c5 -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && timeout 60 python $R/tools/timeline_busy.py $f > $OUT/timeline_c5.txt 2>&1
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
stamp traffic
[ $(left) -gt 150 ] && bash $R/tools/collect_traffic.sh c5 > $OUT/traffic.log 2>&1
cp $R/gpurun_out/r02traffic/c5_*.json $OUT/ 2>/dev/null
stamp lines
for c in target c2 c3 c4; do
  [ $(left) -lt 75 ] && break
  st=3; wu=1; [ $c = c2 ] && st=30 && wu=3
  line $c $st $wu
done
for c in n4k c1; do
  [ $(left) -lt 25 ] && break
  line $c 30 3
done
stamp end
