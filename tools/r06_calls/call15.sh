#!/bin/bash
# round 6, call 15: A/B of the update kernel's tile program with the operand fetches software-pipelined inside a wave
# (csrc/gemm_nt.hip: SGP_KPIPE; make libsthenomi_kpipe.so) -- same bits, one box
cd /root/repo
O=gpurun_out/r06_call15
mkdir -p $O
L=stheno.jl_amd/csrc
cp $L/libsthenomi.so /tmp/lib_default.so
line() {  # tag config
  timeout 600 python bench.py --config $2 --steps 4 --warmup 1 --cpu-sample 0 --no-host-api --no-extras > $O/$2_$1.json 2> $O/$2_$1.err
  python - "$1" "$2" "$O/$2_$1.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[3]).read().strip().split("\n")[-1])
    r = d.get("roofline") or {}
    print("%-8s %-8s step %.2f ms  value %.12g  busy frac %s" % (sys.argv[2], sys.argv[1], d["ms_per_step"], d["logpdf"],
          round(r.get("achieved_while_busy", 0) / 78.6, 3) if r.get("achieved_while_busy") else r.get("frac")))
except Exception as e:
    print(sys.argv[2], sys.argv[1], "FAILED", e)
PY
}
for rep in 1 2; do
  cp /tmp/lib_default.so $L/libsthenomi.so
  for c in c5 target n32k c4; do line default$rep $c; done
  cp $L/libsthenomi_kpipe.so $L/libsthenomi.so
  for c in c5 target n32k c4; do line kpipe$rep $c; done
done
# the variant through the parity / bit-identity suites
timeout 1500 python -m pytest tests/test_gpu_baseline_golden.py tests/test_gpu_parity.py tests/test_gpu_dataflow.py tests/test_gpu_struct_zeros.py tests/test_gpu_fused_potrf.py -m gpu -x -q -p no:cacheprovider > $O/pytest_kpipe.log 2>&1
echo "pytest rc=$?" >> $O/pytest_kpipe.log
grep -E "passed|failed|rc=" $O/pytest_kpipe.log | tail -3
cp /tmp/lib_default.so $L/libsthenomi.so
