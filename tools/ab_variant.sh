#!/bin/bash
# same-box A/B of experiment builds of the library (python -m whisperseg_amd.build --variant TAG -D...) on the GEMM shapes:
#   tools/ab_variant.sh TAG [dtype ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
DTS=${@:-f16m6}
for r in 1 2; do
for v in "" _$TAG; do
  for dt in $DTS; do
    echo "== lib$v $dt run $r"
    WSEG_LIB=$PWD/whisperseg_amd/lib/libwseg$v.so timeout 300 python tools/gemm_bench.py --windows 256 --encoder-only --dtype $dt 2>&1 | grep -v amdgpu.ids | tail -5
    WSEG_LIB=$PWD/whisperseg_amd/lib/libwseg$v.so timeout 300 python tools/gemm_bench.py --rotate 8 --dtype $dt --shapes "4096,3840,1280,0;4096,1280,1280,3;4096,5120,1280,1;4096,1280,5120,3;4096,1280,1280,0" 2>&1 | grep -v amdgpu.ids | tail -5
  done
done
done
