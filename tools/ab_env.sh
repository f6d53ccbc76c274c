#!/bin/bash
# per-kernel rocprofv3 comparison of two environments on tools/quick_bench.py: tools/ab_env.sh "<env A>" "<env B>" [windows]
A=$1; B=$2; W=${3:-256}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$A" "$B"; do
  i=$((i+1)); rm -rf /tmp/abe$i
  export $e
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abe$i -o p -- python3 $ROOT/tools/quick_bench.py --windows $W --iters 2 > /tmp/abe$i.log 2>&1
  unset ${e%%=*}
  grep "iter 1" /tmp/abe$i.log | sed "s/.*ckv [0-9.]* //"
done
python3 - <<PY
import csv, glob
def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6) for r in csv.DictReader(open(f))}
a, b = load("/tmp/abe1"), load("/tmp/abe2")
for k in sorted(set(a) | set(b), key=lambda k: -max(a.get(k, (0, 0))[1], b.get(k, (0, 0))[1]))[:16]:
    ca, ta = a.get(k, (0, 0)); cb, tb = b.get(k, (0, 0))
    print(f"{ta:9.2f} ms ({ca:6d})  {tb:9.2f} ms ({cb:6d})  {k[:100]}")
PY
