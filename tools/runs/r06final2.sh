# r06final2: the whole GPU suite (no -x), smoke
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06final2
mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 5 $O/tests.log | cut -c1-250
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log | cut -c1-300
