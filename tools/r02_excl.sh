#!/bin/bash
cd "$(dirname "$0")/.."
for c in ${EXCL_CFGS:-n4k c2 c3}; do
  for v in ${EXCL_LIST:-0 4096 8192 12288 16384 32768}; do
    SGP_EXCL_MAX=$v timeout 200 python bench.py --config $c --cpu-sample 0 --no-host-api --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', 'EXCL_MAX=$v', 'ms_per_step %.3f' % d['ms_per_step'], 'parity', d.get('parity_rel'))"
  done
done
