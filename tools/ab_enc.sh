#!/bin/bash
# same-box A/B of a variant build against the product library on the whole step (1 024 windows: encoder + cross-K/V + decode):
#   tools/ab_enc.sh TAG [quick_bench args]
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
for rep in 1 2; do for v in product $TAG; do
  if [ $v = product ]; then unset WSEG_LIB; else export WSEG_LIB=$PWD/whisperseg_amd/lib/libwseg_$v.so; fi
  echo "== $v (rep $rep)"
  python tools/quick_bench.py --model large --windows 1024 --dtype f16m6 --iters 3 "$@" 2>&1 | grep "iter [12]"
done; done
