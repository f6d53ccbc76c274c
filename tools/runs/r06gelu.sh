# r06gelu: packed GELU in the 8-column epilogues of the 16-bit / split modes (product) against the scalar form (variant sgelu): encoder GEMMs, whole step, bit-identity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06gelu
mkdir -p $O
for rep in 1 2; do for v in product sgelu; do
  if [ $v = product ]; then unset WSEG_LIB; else export WSEG_LIB=$PWD/whisperseg_amd/lib/libwseg_$v.so; fi
  echo "== $v (rep $rep)" >> $O/ab.txt
  timeout 300 python3 tools/gemm_bench.py --windows 256 --encoder-only --dtype f16x3 2>&1 | grep "M=" >> $O/ab.txt
  timeout 600 python3 tools/quick_bench.py --model large --windows 256 --dtype f16x3 --iters 3 2>&1 | grep "iter [12]" >> $O/ab.txt
done; done
unset WSEG_LIB
cat $O/ab.txt | cut -c1-200
timeout 900 python3 -m pytest tests/test_gemm_gpu.py tests/test_model_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -n 3 $O/tests.log
timeout 900 python3 -m pytest tests/test_parity_sweep_gpu.py -x -q -m gpu -k "f16x3 or trained or f16m6" > $O/sweeps.log 2>&1; echo "sweeps rc=$?"; tail -n 3 $O/sweeps.log
