#!/bin/bash
# round 6, call 11: the fp32 structural zeros -- tests + timing
cd /root/repo
mkdir -p gpurun_out/r06_call11
timeout 900 python -m pytest tests/test_gpu_f32.py tests/test_gpu_struct_zeros.py -m gpu -x -q > gpurun_out/r06_call11/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_call11/pytest.log
tail -5 gpurun_out/r06_call11/pytest.log
timeout 600 python tools/gpu_f32_sz_time.py 4096 10923 21845 > gpurun_out/r06_call11/f32_sz_time.txt 2>&1
cat gpurun_out/r06_call11/f32_sz_time.txt
