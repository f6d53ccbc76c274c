# Decode-step GEMM variants at 1024 rows (256 windows x 4 beams), cold weights (128 rotating copies), per-kernel durations
# from rocprofv3 --kernel-trace --stats.   usage: bash tools/dec_gemm_sweep.sh <out-dir under gpurun_out>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $OUT
S="1024,1280,1280,2;1024,3840,1280,0;1024,5120,1280,1;1024,1280,5120,2"
run() {  # name, env assignments...
  name=$1; shift
  rm -rf /tmp/dgs_$name
  env "$@" true
  ( export "$@" WSEG_DUMMY=1; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dgs_$name -o t -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 100 --rotate 128 --shapes "$S" > $OUT/$name.log 2>&1 )
  f=$(find /tmp/dgs_$name -name "*kernel_stats.csv" | head -1)
  echo "== $name"; grep -E "gemm_h16|splitk_reduce" $f | awk -F'","' '{printf "%-110s calls %s avg_us %.2f\n", substr($1,2,110), $2, $4/1000}'
  cp $f $OUT/$name.kernel_stats.csv
  rm -rf /tmp/dgs_$name
}
run base
run nst3 WSEG_SKINNY_NST=3
run nst4 WSEG_SKINNY_NST=4
run bn128 WSEG_SKINNY_BN128=1
run bn128_nst3 WSEG_SKINNY_BN128=1 WSEG_SKINNY_NST=3
run bn128_nst4 WSEG_SKINNY_BN128=1 WSEG_SKINNY_NST=4
run bn128_nst3_t160 WSEG_SKINNY_BN128=1 WSEG_SKINNY_NST=3 WSEG_SKINNY_TARGET=160
run direct128 WSEG_SKINNY_DIRECT128=1
