cd $GRAFT_REPO_ROOT
O=gpurun_out/r06s
mkdir -p $O
SWEEP_ONLY=sweep4_more python tools/parity_sweep.py --sweeps $O/sweep4_all_modes.json > $O/sweep4.log 2>&1; tail -n 7 $O/sweep4.log | cut -c1-300
python -m pytest tests/test_parity_sweep_gpu.py tests/test_scheduler_gpu.py tests/test_model_gpu.py -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log | cut -c1-220
