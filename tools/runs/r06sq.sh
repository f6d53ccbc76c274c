# r06sq: SQ counters of the 256x256 ping-pong GEMM in the default mode f16x3 on the encoder shapes of a 256-window pass (one rocprofv3 pass per group)
O=$GRAFT_REPO_ROOT/gpurun_out/r06sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SHAPES="128000,3840,1280,0;128000,1280,1280,2;128000,5120,1280,1;128000,1280,5120,2"
n=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  n=$((n+1))
  rm -rf /tmp/sq_$n
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/sq_$n -o pmc -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 1 --dtype f16x3 --shapes "$SHAPES" > $O/sq_$n.log 2>&1
  f=$(find /tmp/sq_$n -name "*counter_collection.csv" | head -n 1)
  python3 - "$f" >> $O/sq_counters.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_h16_pp" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault((r["Kernel_Name"][:70], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in by.items():
    print(k, c, "launches", len(v), "last", v[-1], "mean", sum(v) / len(v))
PY
  tail -n 1 $O/sq_$n.log | cut -c1-200
done
cat $O/sq_counters.txt | cut -c1-220
