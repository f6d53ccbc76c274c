#!/bin/bash
# SQ counters (LDS conflicts, MFMA busy, wait cycles) of the large GEMM on a few shapes; one rocprofv3 pass per counter group.
#   tools/pmc_sq.sh <tag> "<shapes for gemm_bench --shapes>"   ->  gpurun_out/<tag>_sq_<n>.csv
set -u
TAG=${1:-sq}
SHAPES=${2:-"61440,3840,5120,0;61440,3840,1280,0"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS"; do
  n=$((n+1))
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_sq_$n -o pmc -- python3 $ROOT/tools/gemm_bench.py --iters 1 --shapes "$SHAPES" > $OUT/${TAG}_sq_$n.log 2>&1
  find $OUT/${TAG}_sq_$n -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_sq_$n.csv \;
  ls -la $OUT/${TAG}_sq_$n.csv; tail -2 $OUT/${TAG}_sq_$n.log
  rm -rf $OUT/${TAG}_sq_$n
done
