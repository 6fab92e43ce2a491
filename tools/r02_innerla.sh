#!/bin/bash
# A/B of the inner look-ahead (SGP_INNER_LA): K = 128 in-panel updates off the panel stream's serial chain
cd "$(dirname "$0")/.."
for c in ${ILA_CFGS:-c1 n4k c2 c3 c5}; do
  for v in 0 1; do
    SGP_INNER_LA=$v timeout 300 python bench.py --config $c --cpu-sample 0 --no-host-api --steps ${ILA_STEPS:-6} --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$c', 'INNER_LA=$v', 'ms_per_step %.3f' % d['ms_per_step'], 'frac %.3f' % r['frac'], 'parity', d.get('parity_rel'))"
  done
done
