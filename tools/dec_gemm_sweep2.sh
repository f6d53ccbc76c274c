# warm vs cold weights, split targets; per-kernel durations from rocprofv3.  usage: bash tools/dec_gemm_sweep2.sh <out-dir>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
run() {  # name rotate shapes env...
  name=$1; rot=$2; S=$3; shift 3
  rm -rf /tmp/dgs_$name
  ( export "$@" WSEG_DUMMY=1; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dgs_$name -o t -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 100 --rotate $rot --shapes "$S" > $OUT/$name.log 2>&1 )
  f=$(find /tmp/dgs_$name -name "*kernel_stats.csv" | head -1)
  cp $f $OUT/$name.kernel_stats.csv
  rm -rf /tmp/dgs_$name
}
A="1024,1280,1280,2"
B="1024,1280,5120,2"
run a_warm 1 "$A"
run a_cold 128 "$A"
run b_warm 1 "$B"
run b_cold 64 "$B"
run a_cold_t160 128 "$A" WSEG_SKINNY_TARGET=160
run a_cold_t512 128 "$A" WSEG_SKINNY_TARGET=512
run a_warm_t160 1 "$A" WSEG_SKINNY_TARGET=160
run a_warm_t512 1 "$A" WSEG_SKINNY_TARGET=512
run a_cold_bm64 128 "$A" WSEG_SKINNY_BM=64
run a_warm_bm64 1 "$A" WSEG_SKINNY_BM=64
