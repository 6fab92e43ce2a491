#!/bin/bash
# Round-6 evidence, ONE gpurun call -> gpurun_out/r06final (tools/make_r06_summary.py turns it into profiles/r06_*).
# Everything is collected against ONE build: the sha1 of libsthenomi.so is written next to every record and the summary
# tool refuses a collection whose pieces disagree (round-3 verdict: a stale PMC pass had been divided by a new schedule's
# algorithmic bytes).  --kernel-trace / --stats and --pmc are separate runs.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06final
cd $R
SHA=$(sha1sum stheno.jl_amd/csrc/libsthenomi.so | cut -d' ' -f1)
if [ -n "$COLLECT_ONLY_PMC" ]; then
  # add counter passes (PMC_CONFIGS) to an EXISTING collection of the same build; everything else is skipped
  mkdir -p $OUT
  cd /tmp && export TMPDIR=/tmp
else
rm -rf $OUT; mkdir -p $OUT
echo $SHA > $OUT/lib_sha1.txt
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "passed|failed|rc=" $OUT/pytest_gpu.log | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
# ---- bench lines (the default line = c5 + north-star target + size sweep extras)
timeout 900 python $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in c1 n4k c2; do timeout 300 python $R/bench.py --config $c --steps 30 --warmup 3 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
for c in c3 n32k c4 target; do timeout 500 python $R/bench.py --config $c --steps 3 --warmup 1 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; done
# the structured models under the dense schedule (A/B of the structural-zero skipping: same bits)
for c in c3 target; do SGP_STRUCT_ZEROS=0 timeout 500 python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench_${c}_dense.json 2> $OUT/bench_${c}_dense.err; done
# the schedules of round 4 on the same box (SGP_HYBRID=0: serial-deep launches at 65536 columns, the dataflow kernel at 32768)
for c in c5 target n32k c3; do SGP_HYBRID=0 timeout 500 python $R/bench.py --config $c --steps 3 --warmup 1 --cpu-sample 0 --no-host-api --no-extras > $OUT/bench_${c}_nohybrid.json 2> $OUT/bench_${c}_nohybrid.err; done
timeout 400 python $R/bench.py --config c5 --dtype f32 --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench_c5_f32.json 2> $OUT/bench_c5_f32.err
timeout 400 python $R/bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config target --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_target_multi8_loopback.json 2> $OUT/bench_target_multi8_loopback.err
timeout 400 python $R/bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config c5 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_c5_multi8_loopback.json 2> $OUT/bench_c5_multi8_loopback.err
SGP_MULTI_SUBPANEL=0 timeout 400 python $R/bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config c5 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_c5_multi8_loopback_sub0.json 2> $OUT/bench_c5_multi8_loopback_sub0.err
# the launch-based panel chain of rounds 2 - 5 on the same box (SGP_MULTI_PANEL_DF=0: same bits)
for c in c5 target; do SGP_MULTI_PANEL_DF=0 timeout 400 python $R/bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --config $c --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_${c}_multi8_loopback_r5chain.json 2> $OUT/bench_${c}_multi8_loopback_r5chain.err; done
timeout 300 python $R/bench.py --gpus 2 --devices 0,0 --config c4 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_c4_multi2_loopback.json 2> $OUT/bench_c4_multi2_loopback.err
timeout 400 python $R/bench.py --config c5 --force-dist --steps 2 --warmup 1 --cpu-sample 0 > $OUT/bench_c5_dist1.json 2> $OUT/bench_c5_dist1.err
# ---- multi-GPU profiles (serialised per-panel timings) + projections: the default (dataflow panel launches, last look-ahead piece
# fused), the unfused form, the round-5 chain, whole panels -- all on this one box
mprof() {   # tag config P env...
  tag=$1; cfg=$2; PP=$3; shift 3
  env "$@" timeout 600 python $R/tools/gpu_multi_profile.py $cfg $PP $OUT/multi_profile_${cfg}_P${PP}_$tag.json > $OUT/multi_profile_${cfg}_P${PP}_$tag.log 2>&1
  python $R/tools/multi_projection.py $OUT/multi_profile_${cfg}_P${PP}_$tag.json > $OUT/projection_${cfg}_P${PP}_$tag.txt 2>&1
}
for cfg in c5 target; do
  mprof default $cfg 8 X=1
  mprof nofuse $cfg 8 SGP_MULTI_FUSE_LA=0
  mprof r5chain $cfg 8 SGP_MULTI_PANEL_DF=0
  mprof sub0 $cfg 8 SGP_MULTI_SUBPANEL=0
done
mprof nocompact target 8 SGP_MULTI_COMPACT=0
mprof dense target 8 SGP_STRUCT_ZEROS=0
mprof cyclic target 8 SGP_MULTI_OWNERS=cyclic
for P in 2 4; do mprof default c5 $P X=1; done
# ---- small N through sgp_logpdf_batch / concurrent contexts, the mid-N schedule sweep
timeout 600 python $R/tools/gpu_batch_time.py 2048 4096 8192 > $OUT/batch_time.json 2> $OUT/batch_time.err
timeout 900 python $R/tools/gpu_midn_sweep.py 12288 16384 20480 > $OUT/midn_sweep.txt 2> $OUT/midn_sweep.err
# ---- the fp32 instantiation on the structured north-star model, structural zeros skipped and not (same bits)
timeout 400 python $R/tools/gpu_f32_sz_time.py --out $OUT/f32_sz_time.json 4096 10923 21845 > $OUT/f32_sz_time.txt 2>&1
# ---- kernel traces (rocprofv3 --kernel-trace --stats of the same commands)
for c in c5 target c3 c2 n4k c1; do
  st=3; [ $c = c1 ] && st=10
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$c -o $c -- \
      python $R/bench.py --config $c --steps $st --warmup 1 --cpu-sample 0 --no-host-api --no-extras > $OUT/prof_${c}_bench.json 2> $OUT/prof_$c.err
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5_f32 -o c5_f32 -- \
    python $R/bench.py --config c5 --dtype f32 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/prof_c5_f32_bench.json 2> $OUT/prof_c5_f32.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_grad_c2 -o grad_c2 -- \
    python $R/tools/gpu_grad_split.py 16384 > $OUT/prof_grad_c2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c4 -o c4 -- \
    python $R/bench.py --config c4 --steps 2 --warmup 1 --cpu-sample 0 > $OUT/prof_c4_bench.json 2> $OUT/prof_c4.err
# the dominant kernel's rate while it is on the chip, from the trace's begin / end stamps alone (tools/trace_busy.py)
for c in c5 target; do
  tr=$(find $OUT/prof_$c -name "*kernel_trace.csv" | head -1)
  [ -n "$tr" ] && python $R/tools/trace_busy.py $tr $OUT/prof_${c}_bench.json > $OUT/trace_busy_$c.json 2> $OUT/trace_busy_$c.err
  [ -n "$tr" ] && python $R/tools/trace_busy.py $tr $OUT/prof_${c}_bench.json chol_dataflow_fat_kernel > $OUT/trace_busy_${c}_panel_kernel.json 2>> $OUT/trace_busy_$c.err
done
rm -f $OUT/*/*/*kernel_trace.csv $OUT/*/*kernel_trace.csv
fi
# ---- HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the dominant kernel of every line
for c in ${PMC_CONFIGS:-c5 target n4k c2 c3 n32k c4}; do
  for cnt in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $cnt --output-format csv -d $OUT/pmc_${c}_$cnt -o p -- \
        python $R/bench.py --config $c --steps 1 --warmup 0 --cpu-sample 0 --no-host-api --no-extras > $OUT/pmc_${c}_$cnt.bench.json 2> $OUT/pmc_${c}_$cnt.err
    f=$(find $OUT/pmc_${c}_$cnt -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python - "$f" $cnt $SHA > $OUT/pmc_${c}_$cnt.json <<'PY'
import csv, json, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]:
        continue
    a = acc.setdefault(r["Kernel_Name"], [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
print(json.dumps({"counter": sys.argv[2], "lib_sha1": sys.argv[3],
                  "kernels": {k: {"launches": v[0], "sum_KiB": v[1]} for k, v in acc.items() if v[1] > 0}}))
PY
    rm -rf $OUT/pmc_${c}_$cnt
  done
done
if [ -z "$COLLECT_ONLY_PMC" ]; then
# ---- MFMA pipe / clock of the c5 update kernel and the dataflow kernel (c3)
for c in ${MFMA_CONFIGS:-c5 target c3}; do
  timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_${c}_MFMA -o p -- \
      python $R/bench.py --config $c --steps 1 --warmup 0 --cpu-sample 0 --no-host-api --no-extras > /dev/null 2> $OUT/pmc_${c}_MFMA.err
  f=$(find $OUT/pmc_${c}_MFMA -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" $SHA > $OUT/pmc_${c}_MFMA.json <<'PY'
import csv, json, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    a = acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
print(json.dumps({"lib_sha1": sys.argv[2], "kernels": {k: {c: {"launches": v[0], "sum": v[1]} for c, v in d.items()} for k, d in acc.items()}}))
PY
  rm -rf $OUT/pmc_${c}_MFMA
done
for f in $OUT/bench_*.json; do python - $f $SHA <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); d["lib_sha1"] = sys.argv[2]; json.dump(d, open(sys.argv[1], "w"))
except Exception as e:
    print("bad bench file", sys.argv[1], e)
PY
done
ls $OUT | wc -l; head -c 400 $OUT/bench_default.json; echo
fi
