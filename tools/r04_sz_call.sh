cd $GRAFT_REPO_ROOT
for c in c3 target; do for z in 0 1; do echo "== $c SGP_STRUCT_ZEROS=$z"; SGP_STRUCT_ZEROS=$z timeout 600 python bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 --no-extras 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    ln=ln.strip()
    if ln.startswith('{'):
        d=json.loads(ln); print(d['ms_per_step'], d.get('parity_rel'), d.get('logpdf'))
    elif 'rror' in ln: print(ln[:300])
"; done; done
timeout 900 python -m pytest tests/test_gpu_dataflow.py tests/test_gpu_parity.py tests/test_gpu_baseline_golden.py -x -q -m gpu 2>&1 | tail -4
