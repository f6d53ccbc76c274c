# r06final: the whole GPU suite, smoke, the default bench (contract line), and the same bench under rocprofv3 --kernel-trace --stats
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06final
mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 5 $O/tests.log | cut -c1-250
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log; tail -n 2 $O/smoke.log | cut -c1-300
timeout 1200 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_f16x3.csv \;
head -n 6 $O/kernel_stats_f16x3.csv | cut -c1-160; tail -c 300 $O/bench_profiled.json
