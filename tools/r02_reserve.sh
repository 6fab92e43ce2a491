#!/bin/bash
# A/B of the CU reservation for the panel chain (SGP_RESERVE_CU CUs per XCD kept free of trailing-update workgroups)
cd "$(dirname "$0")/.."
for r in ${RESERVE_LIST:-0 -1 1}; do
  echo "== SGP_RESERVE_CU=$r"
  SGP_RESERVE_CU=$r timeout 120 python tools/gpu_potrf_contend.py 2>&1 | grep -v amdgpu.ids | grep running
  for c in ${RESERVE_CFGS:-c2 c3}; do
    SGP_RESERVE_CU=$r timeout 200 python bench.py --config $c --cpu-sample 0 --no-host-api --steps 5 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$c', 'ms_per_step %.3f' % d['ms_per_step'], 'update TF/s %.1f (busy %.1f)' % (r['achieved'], r.get('achieved_while_busy') or 0), 'busy_ms', r.get('busy_ms'), 'parity', d.get('parity_rel'))"
  done
done
