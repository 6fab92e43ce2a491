#!/bin/bash
# round 4, call 1: the XCD-affine task order of the dataflow factorisation -- correctness, then the size / patch sweep
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04a
( timeout 900 python -m pytest tests/test_gpu_dataflow.py -x -q 2>&1 | tail -15 ) > gpurun_out/r04a/pytest_dataflow.txt
( timeout 1500 python tools/gpu_df_order.py 16384 32768 65536 2>&1 ) > gpurun_out/r04a/df_order.txt
tail -5 gpurun_out/r04a/pytest_dataflow.txt
cat gpurun_out/r04a/df_order.txt
