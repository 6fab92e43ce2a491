#!/bin/bash
# Round 5, first GPU call: the new tests, the default bench line (with the grad extra and the new CPU baseline), the
# ownership table A/B (profile + projection + loopback-8) and a kernel split of the gradient.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05a
mkdir -p $OUT; cd $R
sha1sum stheno.jl_amd/csrc/libsthenomi.so | cut -d' ' -f1 > $OUT/lib_sha1.txt
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
for own in balanced cyclic; do
  SGP_MULTI_OWNERS=$own timeout 400 python $R/tools/gpu_multi_profile.py target 8 $OUT/multi_profile_target_P8_$own.json > $OUT/multi_profile_target_P8_$own.log 2>&1
  python $R/tools/multi_projection.py $OUT/multi_profile_target_P8_$own.json > $OUT/projection_target_P8_$own.txt 2>&1
  SGP_MULTI_OWNERS=$own timeout 300 python $R/bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config target --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_target_multi8_$own.json 2> $OUT/bench_target_multi8_$own.err
done
timeout 400 python $R/tools/gpu_multi_profile.py c5 8 $OUT/multi_profile_c5_P8.json > $OUT/multi_profile_c5_P8.log 2>&1
python $R/tools/multi_projection.py $OUT/multi_profile_c5_P8.json > $OUT/projection_c5_P8.txt 2>&1
timeout 300 python $R/bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config c5 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_c5_multi8.json 2> $OUT/bench_c5_multi8.err
# kernel split of the gradient at N = 16384 / 32768
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_grad -o grad -- python $R/tools/gpu_grad_split.py 16384 > $OUT/grad_split_16k.log 2>&1
rm -f $OUT/prof_grad/*/*kernel_trace.csv $OUT/prof_grad/*kernel_trace.csv
timeout 200 python $R/tools/gpu_grad_split.py 32768 > $OUT/grad_split_32k.log 2>&1
grep -h "projected\|ownership" $OUT/projection_*.txt | head -20
head -c 600 $OUT/bench_default.json; echo
