#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02f
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_golden.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
one() { echo -n "$1 $2 " ; env $1 timeout 300 python $R/bench.py --config $2 --steps $3 --warmup 2 --cpu-sample 0 --no-host-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['parity_rel'], round(d['roofline']['frac'],3))"; }
for v in X=1 SGP_STREAM_PRIO=swap SGP_STREAM_PRIO=equal; do one $v c1 30; one $v n4k 30; one $v c2 20; one $v c5 3; done | tee $OUT/prio.txt
