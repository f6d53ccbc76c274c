cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m
mkdir -p $O
timeout 900 python3 bench.py --dtype f16x3 --no-cpu-baseline --no-extra > $O/bench_f16x3.json 2> $O/bench_f16x3.err; tail -c 400 $O/bench_f16x3.json
bash tools/prof_quick.sh r06m_f16x3 --dtype f16x3 --windows 1024 --iters 2 > $O/prof.log 2>&1; tail -n 24 $O/prof.log | cut -c1-200
