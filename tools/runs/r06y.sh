# r06y: the x3 modes on 24-bit block-floating-point cross K / V rows (format 3): correctness tests, every sweep per file, bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y
mkdir -p $O
timeout 900 python3 -m pytest tests/test_model_gpu.py tests/test_scheduler_gpu.py tests/test_large_geometry_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 4 $O/tests.log
timeout 1500 python3 tools/parity_sweep.py --sweeps $O/sweeps_x3.json f16x3 bf16x3 > $O/sweeps_x3.log 2>&1; echo "sweeps rc=$?"; cat $O/sweeps_x3.log | tail -n 14
timeout 900 python3 bench.py --no-extra --no-cpu-baseline > $O/bench_f16x3.json 2> $O/bench_f16x3.err; grep -o '"value": [0-9.]*\|"frac": [0-9.]*\|"decode": [0-9.]*' $O/bench_f16x3.json | head -6; tail -n 3 $O/bench_f16x3.err
