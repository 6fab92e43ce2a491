cd $GRAFT_REPO_ROOT
run() { python bench.py --config $1 --steps 3 --warmup 1 --cpu-sample 0 --no-host-api 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$2', round(d['ms_per_step'],1), 'frac', round(r['frac'],3), 'launches', r['launches'], 'parity', d['parity_rel'])"; }
run c5 default
SGP_WOUT=2048 SGP_WMID=1024 run c5 w2048_m1024
SGP_WOUT=4096 SGP_WMID=1024 run c5 w4096_m1024_rec
SGP_WOUT=8192 SGP_WMID=1024 run c5 w8192_m1024_rec
SGP_WOUT=16384 SGP_WMID=1024 run c5 w16384_m1024_rec
SGP_WOUT=4096 SGP_WMID=512 run c5 w4096_m512_rec
SGP_WOUT=4096 SGP_WMID=1024 run target w4096_m1024_rec
SGP_LOOKAHEAD=0 run c3 serial_default
SGP_LOOKAHEAD=0 SGP_WOUT=4096 SGP_WMID=1024 run c3 serial_w4096_m1024
SGP_LOOKAHEAD=0 SGP_WOUT=2048 SGP_WMID=512 run c3 serial_w2048_m512
SGP_LOOKAHEAD=0 SGP_WOUT=2048 SGP_WMID=512 run c2 serial_w2048_m512
SGP_LOOKAHEAD=0 SGP_WOUT=4096 SGP_WMID=512 run c2 serial_w4096_m512
SGP_LOOKAHEAD=0 run c2 serial_default
