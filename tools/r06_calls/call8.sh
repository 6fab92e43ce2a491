#!/bin/bash
# round 6, call 8: panel groups in the hybrid schedule (far updates once per G panels, K = G x 2048): bit-identity, then timing
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call8
rm -rf $OUT; mkdir -p $OUT
cd $R
for G in 2 3; do
  SGP_HYBRID_GROUP=$G timeout 900 python -m pytest tests/test_gpu_dataflow.py tests/test_gpu_struct_zeros.py tests/test_gpu_baseline_golden.py tests/test_gpu_threads.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_G$G.log 2>&1
  echo "G=$G tests: $(grep -E 'passed|failed' $OUT/pytest_G$G.log | tail -1)"
done
cd /tmp && export TMPDIR=/tmp
for G in 1 2 3 4 1 2; do
  row="G=$G"
  for c in c5 target n32k c3; do
    st=3; [ $c = n32k ] && st=5; [ $c = c3 ] && st=5
    SGP_HYBRID_GROUP=$G timeout 400 python $R/bench.py --config $c --steps $st --warmup 1 --cpu-sample 0 --no-host-api --no-extras > $OUT/bench_${c}_G$G.json 2> $OUT/bench_${c}_G$G.err
    row="$row | $c $(python -c "import json;d=json.load(open('$OUT/bench_${c}_G$G.json'));r=d['roofline'];print('%.1f ms frac %.3f busy %.3f parity %.0e' % (d['ms_per_step'], r['frac'], (r.get('achieved_while_busy') or 0)/78.6, d['parity_rel']))" 2>/dev/null)"
  done
  echo "$row"
done
