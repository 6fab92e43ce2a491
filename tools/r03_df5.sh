#!/bin/bash
mkdir -p gpurun_out
{
SGP_DF_TIMEOUT_S=5 timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -6
for c in c1 n4k c2 c3 n32k; do
  echo "== $c (default policy)"
  timeout 200 python bench.py --config $c --steps 6 --warmup 2 --cpu-sample 0 --no-host-api 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d.get('parity_rel'), r.get('schedule'), round(r['frac'],3), r['kernel'][:40])"
done
echo "== c5 SGP_DATAFLOW=1 (lean at 65536: expected to lose to serial-deep)"
SGP_DATAFLOW=1 timeout 200 python bench.py --config c5 --steps 2 --warmup 1 --cpu-sample 0 --no-host-api 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d.get('parity_rel'), r.get('schedule'), round(r['frac'],3))"
} > gpurun_out/df5.txt 2>&1
cat gpurun_out/df5.txt
