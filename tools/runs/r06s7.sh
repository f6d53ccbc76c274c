# r06s7: sweep 7 (third model, seeds 15000..15249, recorded after the x3 modes' rows were frozen) in every mode, per file; first-step logit
# error of the split modes against the f32 mode at 32 + 32 layers with the shipped rows and (knobs build) with the 24-bit float rows
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s7
mkdir -p $O
L=$PWD/whisperseg_amd/lib
SWEEP_ONLY=sweep7_third_fresh2 timeout 1200 python3 tools/parity_sweep.py --sweeps $O/sweep7.json f32 f16x3 bf16x3 f16m6 f16 bf16 > $O/sweep7.log 2>&1; echo "sweep7 rc=$?"; tail -n 6 $O/sweep7.log
timeout 600 python3 tools/logit_error.py --windows 8 f16x3 bf16x3 > $O/logit_error_shipped.txt 2> $O/logit_error_shipped.err; cat $O/logit_error_shipped.txt | cut -c1-400
WSEG_X3_CKV=k24 WSEG_LIB=$L/libwseg_knobs.so timeout 600 python3 tools/logit_error.py --windows 8 f16x3 bf16x3 > $O/logit_error_k24.txt 2> $O/logit_error_k24.err; cat $O/logit_error_k24.txt | cut -c1-400
WSEG_X3_CKV=f32 WSEG_LIB=$L/libwseg_knobs.so timeout 600 python3 tools/logit_error.py --windows 8 f16x3 > $O/logit_error_f32rows.txt 2> $O/logit_error_f32rows.err; cat $O/logit_error_f32rows.txt | cut -c1-400
