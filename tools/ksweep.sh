cd /tmp && export TMPDIR=/tmp
S="1024,1280,128,2;1024,1280,256,2;1024,1280,640,2;1024,1280,1280,2;1024,1280,2560,2;1024,1280,5120,2"
rm -rf /tmp/ksp; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ksp -o ks -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 50 --rotate 40 --shapes "$S" > /tmp/ks.log 2>&1
f=$(find /tmp/ksp -name "*kernel_trace.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/ksweep_trace.csv; grep custom /tmp/ks.log
