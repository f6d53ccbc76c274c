cd /tmp && export TMPDIR=/tmp
for v in bf16 f16; do
  rm -rf /tmp/abp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp -o ab -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py --model large --windows 256 --iters 2 --dtype $v > /tmp/ab.log 2>&1
  f=$(find /tmp/abp -name "*kernel_stats.csv" | head -1)
  cp $f $GRAFT_REPO_ROOT/gpurun_out/r02k_$v.kernel_stats.csv
  grep 'iter 1' /tmp/ab.log
done
