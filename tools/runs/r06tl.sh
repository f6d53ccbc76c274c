cd $GRAFT_REPO_ROOT
O=gpurun_out/r06tl
mkdir -p $O
L=$PWD/whisperseg_amd/lib
timeout 600 python3 tools/trained_logit_error.py f16x3 bf16x3 f16m6 f16 > $O/shipped.txt 2> $O/shipped.err; cat $O/shipped.txt | cut -c1-300; tail -n 2 $O/shipped.err
for f in k24 f32 bfp; do
WSEG_X3_CKV=$f WSEG_LIB=$L/libwseg_knobs.so timeout 600 python3 tools/trained_logit_error.py f16x3 bf16x3 > $O/rows_$f.txt 2> $O/rows_$f.err; echo "== rows $f"; cat $O/rows_$f.txt | cut -c1-300
done
