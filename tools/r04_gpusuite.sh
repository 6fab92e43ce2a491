#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04suite; mkdir -p $O
( timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $O/pytest_gpu.txt
grep -E "passed|failed|error" $O/pytest_gpu.txt | tail -3
grep -E "^FAILED|^ERROR" $O/pytest_gpu.txt | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
