#!/bin/bash
# prices the L2 locality of the production tile order: the same lower update with the tiles dealt to workgroups at random
# (SCRAMBLE) and, on top, every workgroup starting at its own k offset (ROTATE: desynchronised panel reads)
mkdir -p gpurun_out
export SGP_STAMP_VERBOSE=1
{
for mk in "16384 4096" "32768 2048"; do
  python tools/gpu_gemm_stamps.py $mk
  SGP_STAMP_SCRAMBLE=7919 python tools/gpu_gemm_stamps.py $mk
  SGP_STAMP_SCRAMBLE=7919 SGP_STAMP_ROTATE=1 python tools/gpu_gemm_stamps.py $mk
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/scramble.txt
grep -E "lower|contraction  |launch" gpurun_out/scramble.txt
