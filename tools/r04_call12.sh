#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04i; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_f32.py -q 2>&1 | tail -5 ) | tee $O/pytest_f32.txt | grep -E "passed|failed"
{
for v in "SGP_F32_SERIAL_N=1000000000" "X=1" "SGP_F32_WMID=1024" "SGP_F32_WOUT=2048" "SGP_F32_WOUT=8192 SGP_F32_WMID=1024" "SGP_F32_SERIAL_N=16384"; do
  env $v timeout 600 python tools/gpu_f32_variants.py 16384 32768 65536 2>&1 | grep "N="
done
} | tee $O/f32_variants.txt
