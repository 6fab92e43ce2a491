#!/bin/bash
# round 6, call 13: the ELBO's chunked pipeline (c4) -- in-block halving of the row solve, two solve workgroups per CU, the
# transposition fused with the column sums; every variant on one box
cd /root/repo
O=gpurun_out/r06_call13
mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --config c4 --steps 4 --warmup 1 > $O/c4_$name.json 2> $O/c4_$name.err
  python - "$name" "$O/c4_$name.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
    st = d["stages"]
    print("%-22s step %.2f ms  solve %.2f  red+transpose %.2f  gram %.2f  elbo %.12g parity %.2e" % (
        sys.argv[1], d["ms_per_step"], st["row_solve_ms"], st["reductions_transpose_ms"], st["gram_ms"], d["logpdf"], d["parity_rel"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
}
run old        SGP_X_REC=0 SGP_X_DIV=256 SGP_X_FUSE=0
run rec        SGP_X_REC=1 SGP_X_DIV=256 SGP_X_FUSE=0
run div512     SGP_X_REC=0 SGP_X_DIV=512 SGP_X_FUSE=0
run div1024    SGP_X_REC=0 SGP_X_DIV=1024 SGP_X_FUSE=0
run fuse       SGP_X_REC=0 SGP_X_DIV=256 SGP_X_FUSE=1
run all512     SGP_X_REC=1 SGP_X_DIV=512 SGP_X_FUSE=1
run all512_wb8 SGP_X_REC=1 SGP_X_DIV=512 SGP_X_FUSE=1 SGP_X_WB=8
run all512_wb2 SGP_X_REC=1 SGP_X_DIV=512 SGP_X_FUSE=1 SGP_X_WB=2
run all512_wb16 SGP_X_REC=1 SGP_X_DIV=512 SGP_X_FUSE=1 SGP_X_WB=16
run old2       SGP_X_REC=0 SGP_X_DIV=256 SGP_X_FUSE=0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_struct_zeros.py -m gpu -x -q -k "elbo or vfe or sparse or posterior" > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
