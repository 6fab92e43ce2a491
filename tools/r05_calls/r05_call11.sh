#!/bin/bash
# Round 5: the explicit-posterior path on the device + a flakiness pass (threads / multi suites repeated, both enqueue modes).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05k
mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_examples.py tests/test_gpu_capi_consumer.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_examples.log 2>&1; echo "examples rc=$?"; tail -2 $OUT/pytest_examples.log
for rep in 1 2 3 4 5; do
  timeout 600 python -m pytest tests/test_gpu_threads.py -m gpu -q -x -p no:cacheprovider -s > $OUT/pytest_threads_$rep.log 2>&1; echo "threads rep $rep rc=$?"; grep -h "rerun on the launch-based" $OUT/pytest_threads_$rep.log | tail -1
done
for mt in 0 1; do
  SGP_MULTI_THREADS=$mt timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_multi_threads$mt.log 2>&1; echo "multi SGP_MULTI_THREADS=$mt rc=$?"; tail -1 $OUT/pytest_multi_threads$mt.log
done
