cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g
mkdir -p $O
python -m pytest tests -m gpu -x -q -k "not heldout" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
for w in 1 2 8; do
  echo "== lib windows $w" >> $O/small.txt
  timeout 300 python tools/quick_bench.py --dtype f16m6 --windows $w --iters 4 2>&1 | grep "iter [23]" >> $O/small.txt
done
cat $O/small.txt
timeout 900 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
