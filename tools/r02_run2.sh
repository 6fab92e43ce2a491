#!/bin/bash
# GPU pass 2: new potrf_diag / panel_solve fill / exp + left-looking inner panel A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
one() { # env config steps
  echo -n "$1 $2 " >> $OUT/ab.txt
  env $1 timeout 300 python $R/bench.py --config $2 --steps $3 --warmup 2 --cpu-sample 0 --no-host-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['parity_rel'], round(d['roofline']['frac'],3), round(d['stages']['kernelmatrix_GBps']))" >> $OUT/ab.txt
}
for v in SGP_INNER_LL=0 SGP_INNER_LL=1; do
  for c in c1 n4k c2; do one $v $c 20; done
  for c in c3 c5 target; do one $v $c 3; done
done
one "SGP_INNER_LL=1 SGP_LOOKAHEAD=0" c1 20
one "SGP_INNER_LL=1 SGP_LOOKAHEAD=0" n4k 20
one "SGP_INNER_LL=1 SGP_WOUT=1024" c2 20
one "SGP_INNER_LL=1 SGP_WOUT=2048" c5 3
cat $OUT/ab.txt
for c in c1 c2; do
  SGP_INNER_LL=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o $c -- \
      python $R/bench.py --config $c --steps 5 --warmup 2 --cpu-sample 0 --no-host-api > $OUT/prof_${c}_bench.json 2> $OUT/prof_$c.err
  f=$(find $OUT/prof_$c -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && timeout 60 python $R/tools/timeline_busy.py $f > $OUT/timeline_$c.txt 2>&1
  cat $OUT/timeline_$c.txt
done
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
