#!/bin/bash
# Record a held-out parity sweep (tests/golden/tiny2_sweep.json = sweep2, tiny2_sweep3.json = sweep3, tiny2_sweep4.json = sweep4, tiny3_sweep.json = sweep5 and tiny3_sweep6.json = sweep6, tiny3_sweep7.json = sweep7 on the third model) from the reference in parallel parts:
#   tools/record_sweep.sh sweep2|sweep3|sweep4|sweep5|sweep6|sweep7 [workers=5]
# Every worker runs tools/make_golden.py on ONE thread over its share of the 250 seeds (SWEEP2_PART=lo:hi), the parts are merged in seed
# order (SWEEP2_MERGE=1).  The result is byte-identical to a serial  python tools/make_golden.py --only <sweep>  (same single-thread arithmetic).
set -eu
cd "$(dirname "$0")/.."
NAME=${1:?sweep2 or sweep3}
W=${2:-5}
case $NAME in sweep2) FIRST=5000;; sweep3) FIRST=7000;; sweep4) FIRST=9000;; sweep5) FIRST=11000;; sweep6) FIRST=13000;; sweep7) FIRST=15000;; *) echo "unknown sweep $NAME"; exit 2;; esac
STEP=$(( (250 + W - 1) / W ))
pids=()
for ((i = 0; i < W; i++)); do
  lo=$(( FIRST + i * STEP )); hi=$(( lo + STEP )); [ $hi -gt $(( FIRST + 250 )) ] && hi=$(( FIRST + 250 ))
  [ $lo -ge $hi ] && continue
  OMP_NUM_THREADS=1 MKL_NUM_THREADS=1 SWEEP2_PART=$lo:$hi python tools/make_golden.py --only $NAME > /tmp/record_${NAME}_$lo.log 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
SWEEP2_MERGE=1 OMP_NUM_THREADS=1 python tools/make_golden.py --only $NAME
sha256sum tests/golden/tiny2_sweep*.json tests/golden/tiny2_generate.npz tests/golden/tiny3_* 2>/dev/null || true
