#!/bin/bash
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_dataflow.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15
for c in c1 n4k c2; do
  for df in 0 1; do
    echo "== $c SGP_DATAFLOW=$df"
    SGP_DATAFLOW=$df timeout 300 python bench.py --config $c --steps 10 --warmup 3 --cpu-sample 0 --no-host-api 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('parity_rel'))"
  done
done
} > gpurun_out/df1.txt 2>&1
cat gpurun_out/df1.txt
