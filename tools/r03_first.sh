#!/bin/bash
# First GPU call of the next round (~8 min): what round 2's GPU budget no longer covered, in order of importance.
#   1  the -m gpu suite (serial, as the driver runs it) + smoke
#   2  FETCH_SIZE / WRITE_SIZE passes of the trailing updates under the one-launch-per-panel schedule
#      (bench.py reports roofline.traffic = null for c5 / target until profiles/r02_update_traffic.json is refreshed)
#   3  full bench lines (roofline + cpu_baseline + host_api) of c3, c4, target with the final defaults
#   4  the host-verified tile enumeration of round 2 (tilemap.h) was shipped without a timing: c2 / c3 / c5 lines here
#      against profiles/r02_bench_*.json show its effect
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -3 $OUT/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
bash $R/tools/collect_traffic.sh c5 target > $OUT/traffic.log 2>&1; cp $R/gpurun_out/r02traffic/*.json $OUT/ 2>/dev/null
# (also: move tests/test_host_mirror_on_numpy_double.py::test_differentiation_example_known_answer into the -m gpu suite once it has run here)
for c in c5 target c3 c4 c2 n4k c1; do
  st=3; wu=1; case $c in c2|n4k|c1) st=30; wu=3;; esac
  timeout 300 python $R/bench.py --config $c --steps $st --warmup $wu > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  head -c 300 $OUT/bench_$c.json; echo
done
