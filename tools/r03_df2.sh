#!/bin/bash
mkdir -p gpurun_out
{
ok=1
for a in "128 0 4" "256 128 512" "2048 128 512"; do
  echo "=== df_probe $a"
  timeout 20 tools/df_probe.bin $a 2>&1 | grep -v watchdog | awk 'NR<6 || NR%10==0' | tail -8
  timeout 20 tools/df_probe.bin $a 2>&1 | grep -q "\[done\]" || ok=0
done
echo "probe ok=$ok"
if [ $ok = 1 ]; then
  SGP_DF_TIMEOUT_S=3 timeout 400 python -m pytest tests/test_gpu_dataflow.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15
  for c in c1 n4k c2; do
    for df in 0 1; do
      echo "== $c SGP_DATAFLOW=$df"
      SGP_DF_TIMEOUT_S=3 SGP_DATAFLOW=$df timeout 120 python bench.py --config $c --steps 10 --warmup 3 --cpu-sample 0 --no-host-api 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('parity_rel'))"
    done
  done
fi
} > gpurun_out/df2.txt 2>&1
head -120 gpurun_out/df2.txt
