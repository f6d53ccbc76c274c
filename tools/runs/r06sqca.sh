# r06sqca: SQ counters of the x3 cross-attention kernel (1 024 slots, decode only, 4 generated tokens)
O=$GRAFT_REPO_ROOT/gpurun_out/r06sqca
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
n=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS"; do
  n=$((n+1))
  rm -rf /tmp/sqca_$n
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/sqca_$n -o pmc -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py --dtype f16x3 --windows 1024 --gen 4 --iters 1 --decode-only > $O/sq_$n.log 2>&1
  f=$(find /tmp/sqca_$n -name "*counter_collection.csv" | head -n 1)
  python3 - "$f" >> $O/sq_counters.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dec_cross_attn" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault((r["Kernel_Name"][:70], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in by.items():
    v.sort()
    print(k, c, "launches", len(v), "median", v[len(v) // 2])
PY
  tail -n 1 $O/sq_$n.log | cut -c1-160
done
cat $O/sq_counters.txt | cut -c1-200
