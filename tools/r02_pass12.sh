#!/bin/bash
# GPU pass 12 (~3 min): at N >= 32768, is the serial schedule (no look-ahead: every kernel has the chip to itself) with
# the fused update + potrf_diag launches as fast as the two-stream look-ahead?  Alternating repetitions (boxes drift).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
one() {  # env config steps
  echo -n "$1 $2 "
  env $1 timeout 120 python $R/bench.py --config $2 --steps $3 --warmup 1 --cpu-sample 0 --no-host-api 2>>$OUT/bench_err.log \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(round(d['ms_per_step'],3), d['parity_rel'], r.get('frac'), r.get('achieved_while_busy'))" 2>/dev/null || echo "FAILED"
}
{
for rep in 1 2; do
  one "X=default" c5 3
  one "SGP_LOOKAHEAD=0" c5 3
  one "SGP_LOOKAHEAD=0 SGP_FUSE_POTRF=15" c5 3
done
one "X=default" c3 5
one "SGP_LOOKAHEAD=0 SGP_FUSE_POTRF=15" c3 5
one "SGP_LOOKAHEAD=0" c3 5
one "X=default" target 3
one "SGP_LOOKAHEAD=0 SGP_FUSE_POTRF=15" target 3
one "SGP_LOOKAHEAD=0 SGP_FUSE_POTRF=15 SGP_WOUT=2048" c5 3
one "SGP_LOOKAHEAD=0 SGP_FUSE_POTRF=15 SGP_WOUT=512" c5 3
} | tee $OUT/serial.txt
