#!/bin/bash
# round 6, call 9: the round's final collection (tools/collect_r06.sh) against the final build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p $R/gpurun_out/r06_call9
bash tools/collect_r06.sh > $R/gpurun_out/r06_call9/collect.log 2>&1; tail -6 $R/gpurun_out/r06_call9/collect.log
