#!/bin/bash
# Round 5, fourth GPU call: structural zeros in the gradient (tests + timing on c3 with the skipping on / off).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05d
mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_struct_zeros.py tests/test_gpu_parity.py tests/test_gpu_fused_potrf.py tests/test_gpu_random_programmes.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp; export TMPDIR=/tmp
cat > /tmp/gradc3.py <<'PY'
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import __graft_entry__ as entry, bench_configs as bc
P = entry.load_package()
for name in sys.argv[1:]:
    w = bc.build(P, name)
    P.logpdf(w["fx"], w["y"]); g = P.logpdf_and_gradient(w["fx"], w["y"])
    t0 = time.perf_counter(); P.logpdf(w["fx"], w["y"]); t1 = time.perf_counter()
    g = P.logpdf_and_gradient(w["fx"], w["y"]); t2 = time.perf_counter()
    e, d = P.lib.default_context().factor_work()
    print(f"{name} SGP_STRUCT_ZEROS={os.environ.get('SGP_STRUCT_ZEROS','1')}: logpdf {1e3*(t1-t0):.1f} ms, logpdf+grad {1e3*(t2-t1):.1f} ms, work {e/d:.3f}, "
          f"logpdf {g['logpdf']!r} d_noise {g['noise']!r} d_coef0 {g['terms'][0]['d_coef']!r}", flush=True)
PY
for z in 1 0; do SGP_STRUCT_ZEROS=$z timeout 300 python /tmp/gradc3.py c3 w4k; done 2>&1 | grep -v amdgpu.ids | tee $OUT/grad_c3.txt
