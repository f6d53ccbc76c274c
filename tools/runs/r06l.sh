cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l
mkdir -p $O
L=$PWD/whisperseg_amd/lib
WSEG_X3_CKV=k24 WSEG_LIB=$L/libwseg_knobs.so python tools/parity_sweep.py --sweeps $O/f16m6_k24.json f16m6 > $O/f16m6_k24.log 2>&1; tail -n 3 $O/f16m6_k24.log | cut -c1-300
WSEG_X3_CKV=bfp WSEG_LIB=$L/libwseg_knobs.so python tools/parity_sweep.py --sweeps $O/f16x3_bfp.json f16x3 > $O/f16x3_bfp.log 2>&1; tail -n 3 $O/f16x3_bfp.log | cut -c1-300
