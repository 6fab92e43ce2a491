#!/bin/bash
# round 6, call 10: the sharded gradient with a dense Sigma_y; the multi-GPU test files in full
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call10
rm -rf $OUT; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multi_faults.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" $OUT/pytest.log | tail -8
