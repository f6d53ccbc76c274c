#!/bin/bash
# HBM-side traffic of the decode-step kernels from PMC counters (separate passes, kernel-trace only): per-kernel FETCH_SIZE
# and WRITE_SIZE over tools/quick_bench.py (large geometry, 64 windows so that counter collection stays short).
#   tools/pmc_decode.sh <tag> [windows]  ->  gpurun_out/<tag>_dec_FETCH_SIZE.csv, gpurun_out/<tag>_dec_WRITE_SIZE.csv
set -u
TAG=${1:-r02}
W=${2:-64}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export WSEG_NO_GRAPH=1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_dec_$C -o pmc -- python3 $ROOT/tools/quick_bench.py --windows $W --iters 1 --gen 8 > $OUT/${TAG}_dec_$C.log 2>&1
  find $OUT/${TAG}_dec_$C -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_dec_$C.csv \;
  ls -la $OUT/${TAG}_dec_$C.csv
  rm -rf $OUT/${TAG}_dec_$C
done
