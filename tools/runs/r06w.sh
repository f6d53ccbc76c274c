cd $GRAFT_REPO_ROOT
O=gpurun_out/r06w
mkdir -p $O
L=$PWD/whisperseg_amd/lib
SWEEP_ONLY=sweep5_third_model WSEG_X3_CKV=f32 WSEG_LIB=$L/libwseg_knobs.so python tools/parity_sweep.py --sweeps $O/sweep5_x3_f32ckv.json f16x3 bf16x3 > $O/sweep5_f32ckv.log 2>&1; tail -n 3 $O/sweep5_f32ckv.log | cut -c1-300
