#!/bin/bash
# Round 5, eighth GPU call: the substitution's loads ahead of its stores (panel_solve.h) -- bit-identity suites, same-box A/B
# against the previous build, the chain stamps.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05i
mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_potrf.py tests/test_gpu_dataflow.py tests/test_gpu_parity.py tests/test_gpu_baseline_golden.py tests/test_gpu_f32.py tests/test_gpu_examples.py tests/test_gpu_struct_zeros.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp; export TMPDIR=/tmp
L=$R/stheno.jl_amd/csrc
cp $L/libsthenomi.so /tmp/new.so
for rep in 1 2; do
  for which in new prev; do
    if [ $which = prev ]; then cp $L/libsthenomi_prev.so $L/libsthenomi.so; else cp /tmp/new.so $L/libsthenomi.so; fi
    for c in c1 n4k c2; do
      st=100; [ $c = c2 ] && st=20
      timeout 200 python $R/bench.py --config $c --steps $st --warmup 10 --cpu-sample 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', '$c', 'device-resident', round(d['ms_per_step'],4), 'ms; host API', round(d['host_api']['ms_per_call'],4), 'ms', d['parity_rel'], d['roofline']['schedule'])"
    done
  done
done | tee $OUT/strip44_ab.txt
cp /tmp/new.so $L/libsthenomi.so
SGP_DF_STATS=1 timeout 100 python $R/bench.py --config n4k --steps 2 --warmup 1 --cpu-sample 0 --no-host-api --no-extras 2>&1 | grep -E "chain per column|dataflow n_pad" | tail -2 | tee $OUT/chain_new.txt
