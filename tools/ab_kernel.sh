#!/bin/bash
# Same-box A/B of two builds on one kernel: tools/ab_kernel.sh <kernel-name-substring> [windows]
# expects whisperseg_amd/lib/libwseg_old.so and libwseg_new.so; prints the rocprofv3 average duration of the kernel.
K=$1; W=${2:-256}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for r in 1 2; do for v in old new; do
  cp $ROOT/whisperseg_amd/lib/libwseg_$v.so $ROOT/whisperseg_amd/lib/libwseg.so
  rm -rf /tmp/abp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o ab -- python3 $ROOT/tools/quick_bench.py --model large --windows $W --iters 2 > /tmp/ab.log 2>&1
  f=$(find /tmp/abp -name "*kernel_stats.csv" | head -1)
  echo "$v $(grep "$K" $f | head -1 | awk -F'","|",|,"' '{print $2, $4}') $(grep 'iter 1' /tmp/ab.log | sed 's/.*dec //')"
done; done
