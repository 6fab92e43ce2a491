#!/bin/bash
# round 6, call 9: the round's final collection (tools/collect_r06.sh) against the final build.
# BEFORE the call, HERE: `make -C stheno.jl_amd/csrc all` (tests/test_library_is_built_from_these_sources.py) -- the library
# travels as built.
# The boxes of the pool differ by up to 9 % on the MFMA-bound lines (round 6: c5 1373 ... 1500 ms with one build; the
# latency-bound N = 4096 line is the same everywhere).  PROBE_MAX_C2_MS (optional): the call first times c2 and stops -- one
# minute charged instead of thirteen -- when the box is of the slow kind; every collection that did run is kept and named in
# docs/05_measurement.md.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p $R/gpurun_out/r06_call9
if [ -n "$PROBE_MAX_C2_MS" ]; then
  timeout 300 python bench.py --config c2 --steps 30 --warmup 3 --cpu-sample 0 --no-host-api --no-extras > $R/gpurun_out/r06_call9/probe_c2.json 2> $R/gpurun_out/r06_call9/probe_c2.err
  python - $R/gpurun_out/r06_call9/probe_c2.json $PROBE_MAX_C2_MS <<'PY' || exit 3
import json, sys
ms = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])["ms_per_step"]
print("probe: c2 %.2f ms on this box (limit %s)" % (ms, sys.argv[2]))
sys.exit(0 if ms <= float(sys.argv[2]) else 1)
PY
fi
bash tools/collect_r06.sh > $R/gpurun_out/r06_call9/collect.log 2>&1; tail -6 $R/gpurun_out/r06_call9/collect.log
