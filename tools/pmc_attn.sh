#!/bin/bash
# SQ counters of the encoder attention kernel (tools/enc_attn_bench.py), one rocprofv3 pass per counter group.
set -u
TAG=${1:-attn}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
n=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_$n -o pmc -- python3 $ROOT/tools/enc_attn_bench.py --windows 64 --dtype ${DTYPE:-f16m6} --layers 4 > $OUT/${TAG}_$n.log 2>&1
  find $OUT/${TAG}_$n -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_$n.csv \;
  rm -rf $OUT/${TAG}_$n
done
ls -la $OUT/${TAG}_*.csv
