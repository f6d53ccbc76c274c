#!/bin/bash
# per-kernel rocprofv3 summary of <slots> slots decoding <windows> windows (decode only): tools/prof_slots.sh <slots> <windows> <tag>
S=$1; W=$2; TAG=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o p -- python3 $ROOT/tools/quick_bench.py --windows $W --slots $S --iters 2 --decode-only ${@:4} > /tmp/ps.log 2>&1
tail -1 /tmp/ps.log
f=$(find /tmp/ps -name "*kernel_stats.csv" | head -1)
mkdir -p $ROOT/gpurun_out; cp $f $ROOT/gpurun_out/${TAG}_kernel_stats.csv
