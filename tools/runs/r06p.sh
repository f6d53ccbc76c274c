cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p
mkdir -p $O
python -m pytest tests/test_parity_sweep_gpu.py -m gpu -q --durations=12 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 25 $O/tests.log | cut -c1-220
