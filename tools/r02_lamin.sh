#!/bin/bash
# sweep of SGP_LA_MIN (look-ahead only while more than that many trailing columns remain)
cd "$(dirname "$0")/.."
for c in ${LAMIN_CFGS:-n4k c2 c3}; do
  for v in ${LAMIN_LIST:-0 2048 4096 6144 8192 10240 12288}; do
    SGP_LA_MIN=$v timeout 200 python bench.py --config $c --cpu-sample 0 --no-host-api --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', 'LA_MIN=$v', 'ms_per_step %.3f' % d['ms_per_step'], 'parity', d.get('parity_rel'))"
  done
done
