# r06pmc: HBM traffic of the cross-attention kernel (24-bit block-floating-point rows) from PMC counters, one counter per pass
O=$GRAFT_REPO_ROOT/gpurun_out/r06pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o pmc -- python3 $GRAFT_REPO_ROOT/tools/quick_bench.py --dtype f16x3 --windows 256 --gen 8 --iters 1 --decode-only > $O/pmc_$C.log 2>&1
  f=$(find /tmp/pmc_$C -name "*counter_collection.csv" | head -n 1)
  python3 - "$f" $C > $O/cross_attn_$C.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "dec_cross_attn" in r["Kernel_Name"]]
vals = [float(r["Counter_Value"]) for r in rows]
print(sys.argv[2], "launches", len(vals), "kernel", rows[0]["Kernel_Name"][:90] if rows else None)
if vals:
    vals.sort()
    print("KiB per launch: min %.0f median %.0f max %.0f" % (vals[0], vals[len(vals) // 2], vals[-1]))
PY
  cat $O/cross_attn_$C.txt; tail -n 2 $O/pmc_$C.log
done
