#!/bin/bash
# Round 5, fifth GPU call: the lower-only contraction of the diagonal tiles in the dataflow factorisation (A/B), the new
# structural-zero paths (gradient border, K(z,z)).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05e
mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_dataflow.py tests/test_gpu_struct_zeros.py tests/test_gpu_baseline_golden.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp; export TMPDIR=/tmp
for dl in 1 0 1 0; do
  for c in n4k c2; do
    st=60; [ $c = c2 ] && st=20
    SGP_DF_DIAG_LOWER=$dl timeout 200 python $R/bench.py --config $c --steps $st --warmup 5 --cpu-sample 0 --no-host-api --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('DIAG_LOWER=$dl', '$c', round(d['ms_per_step'],4), 'ms', d['parity_rel'])"
  done
done | tee $OUT/diag_lower_ab.txt
SGP_DF_STATS=1 SGP_DF_DIAG_LOWER=1 timeout 100 python $R/bench.py --config n4k --steps 2 --warmup 1 --cpu-sample 0 --no-host-api --no-extras 2>&1 | grep -E "chain per column|dataflow n_pad" | tail -2 | tee $OUT/chain_lower1.txt
SGP_DF_STATS=1 SGP_DF_DIAG_LOWER=0 timeout 100 python $R/bench.py --config n4k --steps 2 --warmup 1 --cpu-sample 0 --no-host-api --no-extras 2>&1 | grep -E "chain per column|dataflow n_pad" | tail -2 | tee $OUT/chain_lower0.txt
