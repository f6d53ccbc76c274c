cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
L=$PWD/whisperseg_amd/lib
for r in 1 2; do for v in "" _vtold; do
  echo "== lib$v windows 256 run $r" >> $O/vt_ab.txt
  WSEG_LIB=$L/libwseg$v.so timeout 300 python tools/quick_bench.py --dtype f16m6 --windows 256 --iters 3 2>&1 | grep "iter [12]" >> $O/vt_ab.txt
done; done
cat $O/vt_ab.txt
for r in 1 2; do
  WSEG_LIB=$L/libwseg.so timeout 600 python tools/quick_bench.py --dtype f16m6 --windows 1024 --decode-only --iters 3 2>&1 | grep "iter [12]" >> $O/dec1024.txt
done
cat $O/dec1024.txt
timeout 900 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
