cd /tmp && export TMPDIR=/tmp
S="480,1280,1280,2;480,5120,1280,1;480,1280,5120,2;480,3840,1280,0"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/skinny_prof -o sk -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 100 --rotate 40 --shapes "$S" > $GRAFT_REPO_ROOT/gpurun_out/skinny_prof.log 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/skinny_prof -name "*kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/skinny_stats.csv \;
find $GRAFT_REPO_ROOT/gpurun_out/skinny_prof -name "*kernel_trace.csv" -exec cp {} $GRAFT_REPO_ROOT/gpurun_out/skinny_trace.csv \;
rm -rf $GRAFT_REPO_ROOT/gpurun_out/skinny_prof
