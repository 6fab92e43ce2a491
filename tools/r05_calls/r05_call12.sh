#!/bin/bash
# Round 5: the 2-column pair pivot of the 16 x 16 micro-Cholesky (libsthenomi_pair.so, -DSGP_POTRF_PAIR=1) -- suites with the
# variant library, then same-box A/B and the chain stamps.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05l
mkdir -p $OUT; cd $R
L=$R/stheno.jl_amd/csrc
cp $L/libsthenomi.so /tmp/base.so
cp $L/libsthenomi_pair.so $L/libsthenomi.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_golden.py tests/test_gpu_fused_potrf.py tests/test_gpu_dataflow.py tests/test_gpu_f32.py tests/test_gpu_examples.py tests/test_gpu_random_programmes.py tests/test_gpu_struct_zeros.py -m gpu -q -p no:cacheprovider > $OUT/pytest_pair.log 2>&1; echo "pytest (pair) rc=$?"; tail -4 $OUT/pytest_pair.log
cd /tmp; export TMPDIR=/tmp
for rep in 1 2; do
  for which in pair base; do
    if [ $which = base ]; then cp /tmp/base.so $L/libsthenomi.so; else cp $L/libsthenomi_pair.so $L/libsthenomi.so; fi
    for c in c1 n4k c2; do
      st=100; [ $c = c2 ] && st=20
      timeout 200 python $R/bench.py --config $c --steps $st --warmup 10 --cpu-sample 0 --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which', '$c', 'device-resident', round(d['ms_per_step'],4), 'ms; host API', round(d['host_api']['ms_per_call'],4), 'ms', d['parity_rel'], d['roofline']['schedule'])"
    done
  done
done | tee $OUT/pair_ab.txt
cp $L/libsthenomi_pair.so $L/libsthenomi.so
SGP_DF_STATS=1 timeout 100 python $R/bench.py --config n4k --steps 2 --warmup 1 --cpu-sample 0 --no-host-api --no-extras 2>&1 | grep -E "chain per column|dataflow n_pad" | tail -2 | tee $OUT/chain_pair.txt
cp /tmp/base.so $L/libsthenomi.so
