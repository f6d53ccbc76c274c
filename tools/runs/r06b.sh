cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b
mkdir -p $O
python -m pytest tests/test_errors.py tests/test_large_geometry_gpu.py tests/test_scheduler_gpu.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 6 $O/tests.log
L=$PWD/whisperseg_amd/lib
for r in 1 2; do for v in "" _nont; do for w in 8 15 32; do
  echo "== lib$v windows $w run $r" >> $O/nt_ab.txt
  WSEG_LIB=$L/libwseg$v.so timeout 300 python tools/quick_bench.py --dtype f16m6 --windows $w --iters 4 2>&1 | grep "iter [23]" >> $O/nt_ab.txt
done; done; done
cat $O/nt_ab.txt
timeout 600 python tools/gemm_bench.py --windows 256 --encoder-only --dtype f16m6 > $O/gemm_enc.txt 2>&1; tail -n 8 $O/gemm_enc.txt
for r in 1 2; do for b in 340 300; do
  echo "== knobs BIG_MIN_BLOCKS=$b run $r" >> $O/bigmin.txt
  WSEG_BIG_MIN_BLOCKS=$b WSEG_LIB=$L/libwseg_knobs.so timeout 600 python tools/quick_bench.py --dtype f16m6 --windows 1024 --decode-only --iters 3 2>&1 | grep "iter [12]" >> $O/bigmin.txt
done; done
cat $O/bigmin.txt
