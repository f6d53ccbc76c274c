cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h
mkdir -p $O
python -m pytest tests/test_gemm_gpu.py tests/test_bench_gpu.py tests/test_model_gpu.py tests/test_errors.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 15 $O/tests.log | cut -c1-250
L=$PWD/whisperseg_amd/lib
for r in 1 2; do for sx in 64 2; do
  echo "== knobs PP_SPLITK_MAX_S=$sx run $r" >> $O/maxs.txt
  WSEG_PP_SPLITK_MAX_S=$sx WSEG_LIB=$L/libwseg_knobs.so timeout 600 python tools/quick_bench.py --dtype f16m6 --windows 1024 --decode-only --iters 3 2>&1 | grep "iter [12]" >> $O/maxs.txt
done; done
cat $O/maxs.txt
