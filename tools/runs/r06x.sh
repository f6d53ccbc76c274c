cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x
mkdir -p $O
L=$PWD/whisperseg_amd/lib
WSEG_X3_CKV=f32 WSEG_LIB=$L/libwseg_knobs.so timeout 900 python3 bench.py --dtype f16x3 --no-extra --no-cpu-baseline --no-roofline > $O/bench_f32ckv.json 2> $O/bench_f32ckv.err; tail -c 1200 $O/bench_f32ckv.json | head -c 600; echo; grep -o '"value": [0-9.]*\|"window_slots": [0-9]*\|"decode": [0-9.]*' $O/bench_f32ckv.json | head -5; tail -n 3 $O/bench_f32ckv.err
