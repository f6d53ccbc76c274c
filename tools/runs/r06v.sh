cd $GRAFT_REPO_ROOT
O=gpurun_out/r06v
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 5 $O/tests.log | cut -c1-250
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log | cut -c1-300
timeout 1200 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json
