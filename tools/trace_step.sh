#!/bin/bash
# per-dispatch timeline of one decode step (kernel name, duration, gap to the previous kernel's end) from a rocprofv3 kernel trace of
# tools/quick_bench.py --decode-only:   tools/trace_step.sh <slots> <windows> <tag> [quick_bench args]
S=$1; W=$2; TAG=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ts; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ts -o p -- python3 $ROOT/tools/quick_bench.py --windows $W --slots $S --iters 1 --decode-only ${@:4} > /tmp/ts.log 2>&1
tail -1 /tmp/ts.log
f=$(find /tmp/ts -name "*kernel_trace.csv" | head -1)
mkdir -p $ROOT/gpurun_out
python3 - $f $ROOT/gpurun_out/${TAG}_timeline.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void wseg::", "") for r in rows]
# the last beam_step_kernel but 3 starts a decode step in the middle of the last call
idx = [i for i, n in enumerate(names) if n.startswith("wseg::beam_step_kernel") or n.startswith("beam_step_kernel")]
if len(idx) < 6:
    sys.exit("no decode steps in the trace")
a, b = idx[-6], idx[-5]
out = open(sys.argv[2], "w")
prev_end = int(rows[a]["End_Timestamp"])
tot_k = tot_g = 0
for i in range(a + 1, b + 1):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    out.write(f"{names[i][:90]:92s} dur {(e - s) / 1e3:8.2f} us  gap {(s - prev_end) / 1e3:7.2f} us\n")
    tot_k += e - s; tot_g += s - prev_end
    prev_end = e
out.write(f"step: {b - a} kernels, kernel time {tot_k / 1e3:.1f} us, gaps {tot_g / 1e3:.1f} us, wall {(int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3:.1f} us\n")
print(open(sys.argv[2]).read()[-400:])
PY
