#!/bin/bash
# round 6, call 7: the fat dataflow kernel's contraction pipeline -- stages and barrier placement, A/B on one box through variant
# builds of the library (the box's copy of libsthenomi.so is swapped between runs)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call7
rm -rf $OUT; mkdir -p $OUT
cd $R
C=stheno.jl_amd/csrc
cp $C/libsthenomi.so /tmp/libsthenomi_default.so
for v in st2 st3e st4e st3m st4m; do
  cp $C/libsthenomi_$v.so $C/libsthenomi.so
  if [ $v = st3m ] || [ $v = st4m ]; then
    timeout 600 python -m pytest tests/test_gpu_dataflow.py tests/test_gpu_batch.py tests/test_gpu_struct_zeros.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_$v.log 2>&1
    echo "$v tests: $(grep -E 'passed|failed' $OUT/pytest_$v.log | tail -1)"
  fi
  for c in c2 n4k n32k c3; do
    st=10; [ $c = n32k ] && st=4; [ $c = c3 ] && st=4
    ( cd /tmp && timeout 300 python $R/bench.py --config $c --steps $st --warmup 2 --cpu-sample 0 --no-host-api --no-extras > $OUT/bench_${c}_$v.json 2> $OUT/bench_${c}_$v.err )
  done
  ( cd /tmp && timeout 300 python $R/bench.py --config c5 --steps 3 --warmup 1 --cpu-sample 0 --no-host-api --no-extras > $OUT/bench_c5_$v.json 2> $OUT/bench_c5_$v.err )
  timeout 300 python tools/gpu_batch_time.py 4096 8192 > $OUT/batch_$v.json 2> $OUT/batch_$v.err
  python - $OUT $v <<'PY'
import json, sys
out, v = sys.argv[1], sys.argv[2]
row = [v]
for c in ("c2", "n4k", "n32k", "c3", "c5"):
    try:
        d = json.load(open(f"{out}/bench_{c}_{v}.json")); row.append(f"{c} {d['ms_per_step']:.2f} ({d['parity_rel']:.0e})")
    except Exception as e:
        row.append(f"{c} ?")
try:
    b = json.loads(open(f"{out}/batch_{v}.json").read().strip().splitlines()[-1])
    row.append("batch8 n4k %.3f n8k %.3f single n8k %.2f ms" % (b["4096"]["batch"]["8"]["frac"], b["8192"]["batch"]["8"]["frac"], b["8192"]["single_ms"]))
except Exception as e:
    row.append("batch ?")
print(" | ".join(row))
PY
done
cp /tmp/libsthenomi_default.so $C/libsthenomi.so
