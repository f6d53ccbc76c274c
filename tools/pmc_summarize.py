"""Summarise tools/pmc_traffic.sh output into profiles/<tag>_gemm_hbm_traffic.json (bytes per launch of the large GEMM)."""
import csv, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
windows = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
es = 2 if dtype in ("bf16", "f16") else 4      # bytes per logical operand element (split modes: hi | lo rows / M6 rows)
MM = windows * 500
g = [r for r in csv.DictReader(open(f"gpurun_out/{tag}_pmc_FETCH_SIZE.csv")) if "gemm_h16" in r["Kernel_Name"] or "gemm_bf16" in r["Kernel_Name"]]
w = [r for r in csv.DictReader(open(f"gpurun_out/{tag}_pmc_WRITE_SIZE.csv")) if "gemm_h16" in r["Kernel_Name"] or "gemm_bf16" in r["Kernel_Name"]]
g, w = g[-25:], w[-25:]      # (operand preparation of the split modes launches no GEMM; keep the last 5 x 5 launches)
names = ["qkv", "o-proj", "fc1", "fc2", "conv2"]      # dispatch order of tools/gemm_bench.py --encoder-only (3 warm-up + 2 timed each)
shapes = {"qkv": (MM, 3840, 1280, 0), "o-proj": (MM, 1280, 1280, 2), "fc1": (MM, 5120, 1280, 1),
          "fc2": (MM, 1280, 5120, 2), "conv2": (MM, 1280, 3840, 1)}
assert len(g) == len(w) == 25, (len(g), len(w))
out = {}
for i, n in enumerate(names):
    fs = [float(r["Counter_Value"]) for r in g[i * 5:(i + 1) * 5]][2:]
    ws = [float(r["Counter_Value"]) for r in w[i * 5:(i + 1) * 5]][2:]
    M, N, K, epi = shapes[n]
    # operands and plain outputs are 2 bytes; the residual epilogue (epi 2) reads and writes the fp32 residual stream
    alg = (M * K + N * K) * es + M * N * (8 if epi == 2 else es)
    fetch, write = 2 * 1024 * sum(fs) / len(fs), 1024 * sum(ws) / len(ws)
    out[n] = dict(M=M, N=N, K=K, algorithmic_bytes=alg, fetch_bytes=fetch, write_bytes=write, hbm_bytes=fetch + write,
                  ratio=(fetch + write) / alg)
    print(f"{n:7s} algorithmic {alg/1e6:6.0f} MB   fetch(x2) {fetch/1e6:6.0f} MB   write {write/1e6:5.0f} MB   ratio {(fetch+write)/alg:.2f}")
json.dump({"windows": windows, "dtype": dtype, "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace only) on "
                     "tools/gemm_bench.py --windows <windows> --encoder-only; counters are KiB; FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; calibrated in the same run on a 307 MB "
                     "elementwise read that reports 154 MB).  FETCH counts L2 misses, Infinity-Cache hits included.",
           "per_launch": out}, open(f"profiles/{tag}_gemm_hbm_traffic.json", "w"), indent=1)
