#!/bin/bash
# VGPR / AGPR / scratch / LDS / occupancy of every kernel of one source file whose (demangled) name matches a pattern:
#   tools/kernel_resources.sh wseg_gemm.hip 'pp_kernel'
SRC=$1; PAT=${2:-.}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -Rpass-analysis=kernel-resource-usage \
  -c $ROOT/whisperseg_amd/csrc/$SRC -o /tmp/kr_$$.o 2> /tmp/kr_$$.txt
python3 - /tmp/kr_$$.txt "$PAT" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
print("scratch vgpr agpr occ lds  name")
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].split(" [")[0].strip()
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem)
    if not pat.search(dem):
        continue
    f = lambda k: (re.search(k + r": (\d+)", b) or [0, -1])[1]
    print(f(r"ScratchSize \[bytes/lane\]"), f(r" VGPRs"), f(r"AGPRs"), f(r"Occupancy \[waves/SIMD\]"), f(r"LDS Size \[bytes/block\]"), dem[:140])
PY
rm -f /tmp/kr_$$.o /tmp/kr_$$.txt
