cd $GRAFT_REPO_ROOT
O=gpurun_out/r06u
mkdir -p $O
SWEEP_ONLY=sweep5_third_model python tools/parity_sweep.py --sweeps $O/sweep5_all_modes.json > $O/sweep5.log 2>&1; tail -n 7 $O/sweep5.log | cut -c1-300
python -m pytest tests/test_parity_sweep_gpu.py -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 6 $O/tests.log | cut -c1-220
