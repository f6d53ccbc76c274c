#!/bin/bash
# Produce the per-round evidence under gpurun_out/ (run on the GPU box):  tools/profile_round.sh r01
#   <tag>_bench.json              the contract line of `python bench.py` (default workload)
#   <tag>_kernel_stats.csv        rocprofv3 --kernel-trace --stats summary of the SAME workload (timed steps + the live-profiler step;
#                                 no CPU baseline, self-check or extra lines: 5 steps in all)
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python3 bench.py --steps 3 --warmup 1 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-check --no-extra > $OUT/${TAG}_bench_profiled.json 2> $OUT/${TAG}_prof.err
find $OUT/${TAG}_prof -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_kernel_stats.csv \;
head -25 $OUT/${TAG}_kernel_stats.csv
rm -rf $OUT/${TAG}_prof
