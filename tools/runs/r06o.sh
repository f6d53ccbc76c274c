cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 6 $O/tests.log | cut -c1-250
bash tools/profile_round.sh r06x3 > $O/profile_round.log 2>&1; tail -n 2 $O/profile_round.log | cut -c1-200
bash tools/pmc_traffic.sh r06_x3w256 256 f16x3 > $O/pmc.log 2>&1; tail -n 3 $O/pmc.log
