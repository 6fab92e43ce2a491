#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02h
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
one() { echo -n "$1 $2 " ; env $1 timeout 300 python $R/bench.py --config $2 --steps $3 --warmup 3 --cpu-sample 0 --no-host-api 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['parity_rel'], round(d['roofline']['frac'],3))"; }
for c in c1 n4k c2; do one X=1 $c 30; done | tee $OUT/dpp.txt
one X=1 c5 3 | tee -a $OUT/dpp.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c1 -o c1 -- python $R/bench.py --config c1 --steps 10 --warmup 2 --cpu-sample 0 --no-host-api > /dev/null 2>&1
f=$(find $OUT/prof_c1 -name "*kernel_stats.csv" | head -1); head -5 $f
timeout 200 python $R/tools/gpu_illcond.py 2>&1 | grep -v amdgpu | tail -12
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
