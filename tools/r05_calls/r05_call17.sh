#!/bin/bash
# Round 5: hybrid schedule for the gradient path's factorisation (identity border)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05q
mkdir -p $OUT; cd $R; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_dataflow.py tests/test_gpu_struct_zeros.py tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "hybrid or grad or elbo" > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -5 $OUT/pytest_a.log
cd /tmp
{
for hy in 0 1; do
  SGP_HYBRID_GROW=$hy timeout 300 python $R/tools/gpu_grad_split.py 32768 2>&1 | sed "s/^/GROW=$hy /"
  SGP_HYBRID_GROW=$hy timeout 300 python $R/tools/gpu_grad_split.py 24576 2>&1 | sed "s/^/GROW=$hy /"
done
SGP_HYBRID=1 timeout 300 python $R/tools/gpu_grad_split.py 16384 2>&1 | sed "s/^/HYBRID=1 /"
timeout 300 python $R/tools/gpu_grad_split.py 16384 2>&1 | sed "s/^/default /"
} | tee $OUT/grad_hybrid.txt
