#!/bin/bash
# round 6, call 12: the data-sharded ELBO gradient on the multi-GPU context -- its tests + the files around it
cd /root/repo
mkdir -p gpurun_out/r06_call12
timeout 2400 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multi_faults.py tests/test_gpu_parity.py tests/test_gpu_threads.py -m gpu -x -q > gpurun_out/r06_call12/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r06_call12/pytest.log
tail -30 gpurun_out/r06_call12/pytest.log
