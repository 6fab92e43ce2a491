#!/bin/bash
# GPU pass 4: multi-GPU context tests + full suite + assembly v3 + WMID A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -x -q > $OUT/pytest_multi.log 2>&1; echo "rc=$?" >> $OUT/pytest_multi.log
tail -30 $OUT/pytest_multi.log
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
for c in c2 c3 c5 target; do timeout 200 python $R/tools/gpu_assemble_one.py $c 10 2>&1 | tail -1; done | tee $OUT/assemble_times.txt
one() { echo -n "$1 $2 " ; env $1 timeout 300 python $R/bench.py --config $2 --steps $3 --warmup 2 --cpu-sample 0 --no-host-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['parity_rel'], round(d['roofline']['frac'],3), round(d['stages']['kernelmatrix_GBps']), round(d['stages']['assemble_ms'],3))"; }
for v in X=1 SGP_WMID=512 SGP_WMID=256; do for c in c2; do one $v $c 20; done; for c in c3 c5; do one $v $c 3; done; done | tee $OUT/wmid.txt
one "SGP_WMID=512 SGP_WOUT=2048" c5 3 | tee -a $OUT/wmid.txt
one "SGP_WMID=256 SGP_WOUT=512" c2 20 | tee -a $OUT/wmid.txt
