#!/bin/bash
# the production tile program at one workgroup per CU (SGP_STAMP_LDS_PAD) vs two: prices the occupancy a 256 x 128 C tile needs
mkdir -p gpurun_out
export SGP_STAMP_VERBOSE=1
{
for mk in "32768 1024" "32768 4096"; do
  for pad in 0 16384; do
    echo "== lower $mk, dynamic LDS pad $pad"
    SGP_STAMP_LDS_PAD=$pad python tools/gpu_gemm_stamps.py $mk 2>&1 | grep -E "launch |contraction  |whole tile"
  done
done
} > gpurun_out/occupancy.txt 2>&1
cat gpurun_out/occupancy.txt
