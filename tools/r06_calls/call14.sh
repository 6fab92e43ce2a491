#!/bin/bash
# round 6, call 14: the hybrid schedule's two parameters on the final build, one box -- panel width and the panel kernel's
# workgroup count (c5 and the north-star model)
cd /root/repo
O=gpurun_out/r06_call14
mkdir -p $O
run() {  # name config env...
  local name=$1 cfg=$2; shift 2
  env "$@" timeout 600 python bench.py --config $cfg --steps 4 --warmup 1 --cpu-sample 0 --no-host-api --no-extras > $O/${cfg}_$name.json 2> $O/${cfg}_$name.err
  python - "$name" "$cfg" "$O/${cfg}_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().split("\n")[-1])
    print("%-8s %-14s step %.1f ms  logpdf %.12g" % (sys.argv[2], sys.argv[1], d["ms_per_step"], d["logpdf"]))
except Exception as e:
    print(sys.argv[2], sys.argv[1], "FAILED", e)
PY
}
for cfg in c5 target; do
  run default $cfg X=1
  run w1536 $cfg SGP_HYBRID_W=1536
  run w2560 $cfg SGP_HYBRID_W=2560
  run w3072 $cfg SGP_HYBRID_W=3072
  run wgs192 $cfg SGP_HYBRID_WGS=192
  run wgs224 $cfg SGP_HYBRID_WGS=224
  run wgs128 $cfg SGP_HYBRID_WGS=128
  run default2 $cfg X=1
done
