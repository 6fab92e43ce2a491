#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02i
mkdir -p $OUT
cd $R
python tools/gpu_potrf_phases.py 2>&1 | grep -v amdgpu | tee $OUT/potrf_phases.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_golden.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
one() { echo -n "$1 $2 " ; env $1 timeout 300 python $R/bench.py --config $2 --steps $3 --warmup 3 --cpu-sample 0 --no-host-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['parity_rel'], round(d['roofline']['frac'],3))"; }
for c in c1 n4k c2; do one X=1 $c 30; done | tee $OUT/t.txt
