#!/bin/bash
# GPU pass 11 (one gpurun call, ~4 min): the new defaults (fused launches below 32768, one panel up to 4096, panel wave
# priority 3) through the whole -m gpu suite; A/B of the one-Newton-step pivot chain (libsthenomi_n1.so swapped in);
# panel width at N = 8192; fused launches at c3; profiles of c1 / n4k / c2 with the new defaults.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02g
mkdir -p $OUT
T0=$(date +%s)
LIMIT=${LIMIT:-400}
left() { echo $(( LIMIT - ( $(date +%s) - T0 ) )); }
stamp() { echo "== $1 at $(( $(date +%s) - T0 )) s" | tee -a $OUT/progress.txt; }
CS=$R/stheno.jl_amd/csrc
one() {  # env config steps
  echo -n "$1 $2 "
  env $1 timeout 120 python $R/bench.py --config $2 --steps $3 --warmup 3 --cpu-sample 0 --no-host-api 2>>$OUT/bench_err.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(round(d['ms_per_step'],4), d['parity_rel'], r.get('frac'))" 2>/dev/null || echo "FAILED"
}
stamp suite
cd $R
timeout 200 python -m pytest tests -m gpu -q -n 4 --dist loadfile -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
stamp defaults
{ one "X=0" c1 30; one "X=0" n4k 30; one "X=0" c2 15; } | tee $OUT/defaults.txt
stamp newton1
cp $CS/libsthenomi.so $CS/libsthenomi_n2.so && cp $CS/libsthenomi_n1.so $CS/libsthenomi.so
{ one "X=n1" c1 30; one "X=n1" n4k 30; one "X=n1" c2 15; } | tee $OUT/newton1.txt
cd $R
timeout 200 python -m pytest tests -m gpu -q -n 4 --dist loadfile -p no:cacheprovider > $OUT/pytest_gpu_n1.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_n1.log
tail -4 $OUT/pytest_gpu_n1.log
timeout 60 python $R/tools/gpu_illcond_err.py > $OUT/illcond_n1.log 2>&1
cd /tmp
cp $CS/libsthenomi_n2.so $CS/libsthenomi.so
timeout 60 python $R/tools/gpu_illcond_err.py > $OUT/illcond_n2.log 2>&1; paste $OUT/illcond_n2.log $OUT/illcond_n1.log | cut -c1-200
stamp n8192
for w in 0 2048 8192; do echo -n "SGP_WOUT=$w "; SGP_WOUT=$w timeout 60 python $R/tools/gpu_bign.py 8192 2>&1 | grep "N="; done | tee $OUT/n8192.txt
stamp knobs
{
  one "SGP_FUSE_MAX_N=65536" c3 5
  one "X=0" c3 5
  one "SGP_PS_DIV=128" c2 15
  one "SGP_PS_DIV=128" c3 5
  one "SGP_PANEL_PRIO=0" c3 5
  one "X=0" c5 3
  one "SGP_PANEL_PRIO=0" c5 3
} | tee $OUT/knobs.txt
stamp profiles
for c in c2 n4k c1; do
  [ $(left) -lt 60 ] && break
  st=3; [ $c = c1 ] && st=10; [ $c = n4k ] && st=10
  timeout 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o $c -- \
      python $R/bench.py --config $c --steps $st --warmup 2 --cpu-sample 0 --no-host-api > $OUT/prof_${c}_bench.json 2> $OUT/prof_$c.err
  f=$(find $OUT/prof_$c -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && timeout 30 python $R/tools/timeline_busy.py $f > $OUT/timeline_$c.txt 2>&1
done
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
for c in c1 n4k c2; do
  [ $(left) -lt 30 ] && break
  timeout 90 python $R/bench.py --config $c --steps 30 --warmup 3 --cpu-sample 8192 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
done
stamp end
