cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k
mkdir -p $O
python tools/parity_sweep.py --sweeps $O/r06_parity_sweeps.json > $O/sweeps.log 2>&1; echo "rc=$?" >> $O/sweeps.log; tail -n 14 $O/sweeps.log | cut -c1-300
python -m pytest tests/test_parity_sweep_gpu.py -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 12 $O/tests.log | cut -c1-300
