#!/bin/bash
# round 6, call 16: the clock the chip holds under the update kernel, and the MFMA pipe's share of those cycles, from ONE rocprofv3
# run with counters + kernel trace (one stream: the launches alone on the chip) -- tools/pmc_clock.py
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r06_call16
mkdir -p $O
for c in c5 c4; do
  SGP_HYBRID_SERIAL=1 timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_$c -o p -- \
      python $R/bench.py --config $c --steps 1 --warmup 0 --cpu-sample 0 --no-host-api --no-extras > $O/bench_$c.json 2> $O/bench_$c.err
  k="gemm_nt_dma_kernel<1>"; [ $c = c4 ] && k="gemm_nt_dma_kernel<0>"
  python $R/tools/pmc_clock.py $O/pmc_$c "$k" > $O/clock_$c.json 2> $O/clock_$c.err
  cat $O/clock_$c.json $O/clock_$c.err
  python $R/tools/pmc_clock.py $O/pmc_$c "chol_dataflow_fat_kernel" > $O/clock_${c}_panel.json 2>/dev/null
  rm -rf $O/pmc_$c
done
