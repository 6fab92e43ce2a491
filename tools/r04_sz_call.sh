cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_struct_zeros.py tests/test_gpu_fused_potrf.py -x -q -m gpu 2>&1 | tail -3
for c in target; do for z in 1; do SGP_STRUCT_ZEROS=$z timeout 600 python bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 --no-extras 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    ln=ln.strip()
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline']; print('$c', d['ms_per_step'], d.get('parity_rel'), 'roofline', r['frac'], r['avg_launch_ms'])
"; done; done
SGP_DATAFLOW=0 python tools/gpu_df_variants.py 32768 dense:SGP_STRUCT_ZEROS=0 2>&1 | grep best | cut -c1-60
