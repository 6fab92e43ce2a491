#!/bin/bash
# Round 5, sixth GPU call: lower-only tile (0, 0) in the fused inner update of the launch-based schedule (c1 / N <= 3072) -- A/B
# against the previous library build (libsthenomi_prev.so travels with the snapshot), plus the bit-identity suites.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05f
mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_potrf.py tests/test_gpu_dataflow.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp; export TMPDIR=/tmp
L=$R/stheno.jl_amd/csrc
cp $L/libsthenomi.so /tmp/new.so
for rep in 1 2; do
  for which in new prev; do
    if [ $which = prev ]; then cp $L/libsthenomi_prev.so $L/libsthenomi.so; else cp /tmp/new.so $L/libsthenomi.so; fi
    for c in c1 n4k; do
      timeout 200 python $R/bench.py --config $c --steps 100 --warmup 10 --cpu-sample 0 --no-host-api --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', '$c', round(d['ms_per_step'],4), 'ms', d['parity_rel'], d['roofline']['schedule'])"
    done
    SGP_DATAFLOW=0 timeout 200 python $R/bench.py --config n4k --steps 100 --warmup 10 --cpu-sample 0 --no-host-api --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', 'n4k launches', round(d['ms_per_step'],4), 'ms', d['parity_rel'], d['roofline']['schedule'])"
  done
done | tee $OUT/lower00_ab.txt
cp /tmp/new.so $L/libsthenomi.so
