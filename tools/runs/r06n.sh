cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n
mkdir -p $O
L=$PWD/whisperseg_amd/lib
SWEEP_ONLY=sweep3_fresh python tools/parity_sweep.py --sweeps $O/sweep3_all_modes.json > $O/sweep3.log 2>&1; tail -n 7 $O/sweep3.log | cut -c1-300
SWEEP_ONLY=sweep3_fresh WSEG_X3_CKV=k24 WSEG_LIB=$L/libwseg_knobs.so python tools/parity_sweep.py --sweeps $O/sweep3_f16m6_k24.json f16m6 > $O/sweep3_k24.log 2>&1; tail -n 2 $O/sweep3_k24.log | cut -c1-300
