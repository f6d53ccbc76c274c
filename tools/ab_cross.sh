#!/bin/bash
# same-box A/B of cross-attention variants on the decode step (1 024 slots, encoder states precomputed): product library, the build without
# the K / V prefetch ahead of the serial sections (--variant nopf -DWSEG_BFP_PREFETCH=0) and the 24-bit format of r03-r04 (knob build, WSEG_X3_CKV=k24)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for v in product nopf k24; do
    if [ $v = product ]; then unset WSEG_LIB; else export WSEG_LIB=$PWD/whisperseg_amd/lib/libwseg_$v.so; fi
    if [ $v = k24 ]; then export WSEG_X3_CKV=k24; else unset WSEG_X3_CKV; fi
    echo "== $v (rep $rep)"
    python tools/quick_bench.py --model large --windows 1024 --dtype f16m6 --iters 3 --decode-only 2>&1 | grep "iter [12]"
  done
done
