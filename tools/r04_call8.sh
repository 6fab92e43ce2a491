#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04h; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q -k "elbo or sparse or vfe or titsias or golden" 2>&1 | tail -15 ) > $O/pytest_elbo.txt
grep -E "passed|failed" $O/pytest_elbo.txt | tail -2; grep -E "^FAILED" $O/pytest_elbo.txt | head
for v in "base SGP_VFE_OVERLAP=0 SGP_SPLITK_SUB=0" "sub SGP_VFE_OVERLAP=0" "overlap SGP_SPLITK_SUB=0" "both X=1"; do
  set -- $v; tag=$1; shift
  env "$@" timeout 600 python bench.py --config c4 --steps 5 --warmup 2 --cpu-sample 0 > $O/bench_c4_$tag.json 2> $O/bench_c4_$tag.err
  python -c "
import json; d=json.load(open('$O/bench_c4_$tag.json')); print('$tag', 'ms_per_step', round(d['ms_per_step'],2), 'parity', d['parity_rel'], {k: round(v,1) for k,v in (d['stages'] or {}).items()})" || tail -3 $O/bench_c4_$tag.err
done
