#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun).
# --kernel-trace/--stats and --pmc are separate runs (never combined with other trace domains).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -o c5 -- \
    python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 > $OUT/prof_c5_bench.json 2> $OUT/prof_c5.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- \
    python $R/bench.py --config n32k --steps 1 --warmup 0 --cpu-sample 0 > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- \
    python $R/bench.py --config n32k --steps 1 --warmup 0 --cpu-sample 0 > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -o s -- \
    python $R/bench.py --config n32k --steps 1 --warmup 0 --cpu-sample 0 > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
ls -R $OUT/prof_c5 $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq | head -30
cat $OUT/prof_c5_bench.json | cut -c1-300
