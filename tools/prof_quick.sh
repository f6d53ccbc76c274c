#!/bin/bash
# rocprofv3 per-kernel summary of tools/quick_bench.py:  tools/prof_quick.sh <tag> [quick_bench args...]
# -> gpurun_out/<tag>_kernel_stats.csv (+ a compact top-30 listing on stdout)
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq_$TAG -o p -- python3 $ROOT/tools/quick_bench.py "$@" > /tmp/pq_$TAG.log 2>&1
tail -2 /tmp/pq_$TAG.log
f=$(find /tmp/pq_$TAG -name "*kernel_stats.csv" | head -1)
mkdir -p $ROOT/gpurun_out
cp "$f" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
python3 - "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:30]:
    name = re.sub(r"\(.*", "", r["Name"])[:110]
    print("%8d calls %10.1f us avg %9.2f ms total %5.1f%%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"]), name))
PY
