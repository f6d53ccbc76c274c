#!/bin/bash
# same-box A/B of the one-wave-per-SIMD GEMM kernel (gemm_w4_kernel) against the ping-pong kernel, from the knob build of the library:
#   python -m whisperseg_amd.build --variant w4 -DWSEG_KNOBS=1 ; tools/ab_w4.sh [windows]
cd ${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-256}
export WSEG_LIB=$PWD/whisperseg_amd/lib/libwseg_w4.so
for dt in bf16 f16m6; do
  for w4 in 0 1; do
    echo "== dtype $dt  WSEG_GEMM_W4=$w4"
    WSEG_GEMM_W4=$w4 python tools/gemm_bench.py --windows $W --encoder-only --dtype $dt --iters 10 2>&1 | grep -v "^$"
  done
done
