cd /tmp; export TMPDIR=/tmp
for z in 1 0; do
SGP_STRUCT_ZEROS=$z rocprofv3 --kernel-trace --output-format csv -d /tmp/szprof$z -o t -- python $GRAFT_REPO_ROOT/bench.py --config target --steps 1 --warmup 0 --cpu-sample 0 --no-host-api --no-extras > /dev/null 2>&1
f=$(find /tmp/szprof$z -name "*kernel_trace.csv" | head -1)
python - "$f" $z <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
big=[r for r in rows if 'gemm_nt_dma_potrf_kernel<1, true>' in r['Kernel_Name']]
# the last step's 63 launches
big=big[-63:]
out=[]
for r in big:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6
    g=int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',0))//512
    out.append((g,round(d,2)))
print("SZ",sys.argv[2], "sum", round(sum(d for _,d in out),1)); print(out)
PY
done
