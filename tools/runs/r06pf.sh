# r06pf: A/B of the first-V-rows prefetch + LDS row scales in the x3 cross-attention kernel (product) against the variant without (nopf)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06pf
mkdir -p $O
for rep in 1 2; do for v in product nopf; do
  if [ $v = product ]; then unset WSEG_LIB; else export WSEG_LIB=$PWD/whisperseg_amd/lib/libwseg_$v.so; fi
  echo "== $v (rep $rep)" >> $O/ab.txt
  timeout 600 python3 tools/quick_bench.py --model large --windows 1024 --dtype f16x3 --iters 3 --decode-only 2>&1 | grep "iter [12]" >> $O/ab.txt
done; done
unset WSEG_LIB
cat $O/ab.txt
timeout 1200 python3 -m pytest tests/test_model_gpu.py tests/test_scheduler_gpu.py tests/test_large_geometry_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
timeout 900 python3 -m pytest tests/test_parity_sweep_gpu.py -x -q -m gpu -k "f16x3 or trained" > $O/sweeps.log 2>&1; echo "sweeps rc=$?"; tail -n 3 $O/sweeps.log
