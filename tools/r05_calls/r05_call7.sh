#!/bin/bash
# Round 5, seventh GPU call: same-box A/B of the small / mid sizes against the round-4 library (libsthenomi_r04.so, built from
# commit cd4d67c) -- boxes of the pool differ by up to 1.7 x at small N, so only pairs from one box say anything.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05g
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
L=$R/stheno.jl_amd/csrc
cp $L/libsthenomi.so /tmp/new.so
for rep in 1 2; do
  for which in r05 r04; do
    if [ $which = r04 ]; then cp $L/libsthenomi_r04.so $L/libsthenomi.so; else cp /tmp/new.so $L/libsthenomi.so; fi
    for c in c1 n4k c2; do
      st=100; [ $c = c2 ] && st=20
      SGP_ALLOW_MISSING_SYMBOLS=1 timeout 200 python $R/bench.py --config $c --steps $st --warmup 10 --cpu-sample 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', '$c', 'device-resident', round(d['ms_per_step'],4), 'ms; host API', round(d['host_api']['ms_per_call'],4), 'ms', d['parity_rel'], d['roofline']['schedule'])"
    done
  done
done | tee $OUT/r04_vs_r05_small.txt
cp /tmp/new.so $L/libsthenomi.so
rocm-smi --showclocks 2>/dev/null | head -20 > $OUT/clocks.txt
