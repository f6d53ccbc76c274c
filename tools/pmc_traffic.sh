#!/bin/bash
# HBM traffic of the dominant GEMM kernel from PMC counters, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit in
# one pass), kernel-trace only.  Runs tools/gemm_bench.py on the encoder shapes of the bench workload (default 256 windows) with few
# iterations, because counter collection serialises every dispatch:
#   tools/pmc_traffic.sh <tag> [windows] [dtype]   ->  gpurun_out/<tag>_pmc_FETCH_SIZE.csv, gpurun_out/<tag>_pmc_WRITE_SIZE.csv
set -u
TAG=${1:-r01}
WINDOWS=${2:-256}
DTYPE=${3:-bf16}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_$C -o pmc -- python3 $ROOT/tools/gemm_bench.py --windows $WINDOWS --iters 2 --encoder-only --dtype $DTYPE > $OUT/${TAG}_pmc_$C.log 2>&1
  find $OUT/${TAG}_pmc_$C -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_pmc_$C.csv \;
  ls -la $OUT/${TAG}_pmc_$C.csv
  rm -rf $OUT/${TAG}_pmc_$C
done
