#!/bin/bash
# Round 3, first GPU call (reduced tools/r03_first.sh): -m gpu suite + smoke, FETCH/WRITE passes of the trailing updates
# under the serial schedule (c5, target), and quick step times of c5 / target / c2 / n4k / c1 with HEAD.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
bash $R/tools/collect_traffic.sh c5 target > $OUT/traffic.log 2>&1; cp $R/gpurun_out/r02traffic/*.json $OUT/ 2>/dev/null
for c in c5 target c2 n4k c1; do
  st=3; wu=1; case $c in c2|n4k|c1) st=30; wu=3;; esac
  timeout 300 python $R/bench.py --config $c --steps $st --warmup $wu --cpu-sample 0 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  head -c 200 $OUT/bench_$c.json; echo
done
