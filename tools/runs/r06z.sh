# r06z: the fresh sweep 6 (third model, seeds 13000..13249) in every mode, per file; then the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z
mkdir -p $O
SWEEP_ONLY=sweep6_third_fresh timeout 1200 python3 tools/parity_sweep.py --sweeps $O/sweep6.json f32 f16x3 bf16x3 f16m6 f16 bf16 > $O/sweep6.log 2>&1; echo "sweep6 rc=$?"; tail -n 7 $O/sweep6.log
