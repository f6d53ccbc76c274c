cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tj; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tj -o p -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py --dtype f16m6 --windows 1024 --iters 2 > /tmp/tj.log 2>&1
tail -n 2 /tmp/tj.log
f=$(find /tmp/tj -name "*kernel_trace.csv" | head -1)
python3 - $f > $GRAFT_REPO_ROOT/gpurun_out/r06j_copies.txt <<'PY'
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void wseg::", "")[:60] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
ctx = collections.Counter()
big = []
for i, n in enumerate(names):
    if "copyBuffer" in n and "Rect" not in n:
        key = (names[i - 1] if i else "-", names[i + 1] if i + 1 < len(names) else "-", "big" if dur[i] > 50 else "small")
        ctx[key] += 1
        if dur[i] > 50: big.append((i, dur[i], rows[i].get("Grid_Size"), rows[i].get("Workgroup_Size")))
for k, v in ctx.most_common(25): print(v, k)
print("big copies:", len(big), "total us", sum(b[1] for b in big))
for b in big[:40]: print(b, names[b[0]-2:b[0]+3])
PY
tail -n 60 $GRAFT_REPO_ROOT/gpurun_out/r06j_copies.txt | cut -c1-260
