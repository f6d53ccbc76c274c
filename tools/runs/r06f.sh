cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f
mkdir -p $O
L=$PWD/whisperseg_amd/lib
python -m pytest tests/test_large_geometry_gpu.py tests/test_scheduler_gpu.py tests/test_model_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
WSEG_LIB=$L/libwseg_preea.so python -m pytest tests/test_large_geometry_gpu.py -m gpu -x -q -k "admission" > $O/tests_old_lib.log 2>&1; echo "rc=$?" >> $O/tests_old_lib.log; tail -n 12 $O/tests_old_lib.log | cut -c1-200
for v in "" _preea; do for w in 1 2 8; do
  echo "== lib$v windows $w" >> $O/small.txt
  WSEG_LIB=$L/libwseg$v.so timeout 300 python tools/quick_bench.py --dtype f16m6 --windows $w --iters 4 2>&1 | grep "iter [23]" >> $O/small.txt
done; done
cat $O/small.txt
