#!/bin/bash
# L2 / L1 / address-unit counters of the large GEMM (is the K loop bound by operand delivery?); one rocprofv3 pass per group.
#   tools/pmc_l2.sh <tag> <dtype> "<shapes>"   ->  gpurun_out/<tag>_l2_<n>.csv
set -u
TAG=${1:-l2}; DT=${2:-f16m6}
SHAPES=${3:-"128000,3840,1280,0"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for C in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_BUSY_avr TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  n=$((n+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_l2_$n -o pmc -- python3 $ROOT/tools/gemm_bench.py --iters 1 --dtype $DT --shapes "$SHAPES" > $OUT/${TAG}_l2_$n.log 2>&1
  find $OUT/${TAG}_l2_$n -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_l2_$n.csv \;
  echo "== $C"; python3 - $OUT/${TAG}_l2_$n.csv <<'PY'
import csv, sys, collections
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no csv", e); sys.exit(0)
acc = collections.defaultdict(list)
for r in rows:
    if "gemm_h16" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "n=%d mean=%.4g" % (len(v), sum(v) / len(v)))
PY
  tail -1 $OUT/${TAG}_l2_$n.log
  rm -rf $OUT/${TAG}_l2_$n
done
