cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d
mkdir -p $O
python -m pytest tests/test_model_gpu.py tests/test_large_geometry_gpu.py tests/test_parity_sweep_gpu.py tests/test_segment_gpu.py -m gpu -x -q -k "not heldout" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
L=$PWD/whisperseg_amd/lib
for r in 1 2; do for v in "" _preea; do
  echo "== lib$v windows 256 run $r" >> $O/ea_ab.txt
  WSEG_LIB=$L/libwseg$v.so timeout 300 python tools/quick_bench.py --dtype f16m6 --windows 256 --iters 3 2>&1 | grep "iter [12]" >> $O/ea_ab.txt
done; done
cat $O/ea_ab.txt
for gm in 4 2 6 8; do
  echo "== knobs GROUP_M=$gm" >> $O/gm.txt
  WSEG_GEMM_GROUP_M=$gm WSEG_LIB=$L/libwseg_knobs.so timeout 300 python tools/gemm_bench.py --windows 256 --encoder-only --dtype f16m6 2>&1 | grep "M=" >> $O/gm.txt
done
cat $O/gm.txt
