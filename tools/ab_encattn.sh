#!/bin/bash
# same-box A/B of an encoder-attention variant build against the product library: per-kernel rocprofv3 average on an 8-layer encoder
#   tools/ab_encattn.sh TAG      (python -m whisperseg_amd.build --variant TAG -D...)
cd ${GRAFT_REPO_ROOT:-/root/repo}
ROOT=$PWD; TAG=$1
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in product $TAG; do
  if [ $v = product ]; then unset WSEG_LIB; else export WSEG_LIB=$ROOT/whisperseg_amd/lib/libwseg_$v.so; fi
  rm -rf /tmp/ea_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ea_$v -o p -- python3 $ROOT/tools/enc_attn_bench.py --dtype f16m6 --layers 8 > /tmp/ea_$v.log 2>&1
  f=$(find /tmp/ea_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v (rep $rep): $(tail -1 /tmp/ea_$v.log)"; grep "enc_attention" $f | awk -F, '{printf "   %s calls  %.1f us avg\n", $(NF-6), $(NF-4)/1e3}'
done; done
