"""Build libwseg.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m whisperseg_amd.build [--force] [--stamps N] [--variant TAG -DNAME=V ...]

The shared library lands in whisperseg_amd/lib/libwseg.so; it is git-ignored but travels with the
gpurun snapshot.  hipcc cross-compiles without a GPU.

Every source is compiled with -save-temps: the device assembly hipcc leaves behind is digested by tools/isa_lint.py into
build/<name>.lint.json (per kernel: VGPRs, AGPRs, SGPRs, LDS, scratch bytes, spill counts; hazards around inline-asm MFMAs) and
deleted; tests/test_isa_lint.py asserts on the digests, i.e. on the very objects the library is linked from.

--stamps N builds a SECOND library, lib/libwseg_stamps<N>.so, with -DWSEG_STAMPS=N: decode kernel N (1 self-attention,
2 packed cross-attention, 3 24-bit cross-attention) records s_memrealtime stamps at its phase boundaries (tools/stamps.py
loads it through WSEG_LIB).  The product library never carries them.
--variant TAG -D... builds lib/libwseg_<TAG>.so with the extra defines: A/B timing of an experiment knob against the product
library on one box (WSEG_LIB=whisperseg_amd/lib/libwseg_<TAG>.so python tools/gemm_bench.py ...).
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libwseg.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=off"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True, stamps=0, variant="", defines=()):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build" if not stamps else "build/stamps%d" % stamps)
    lib = LIB if not stamps else os.path.join(LIBDIR, "libwseg_stamps%d.so" % stamps)
    flags = FLAGS + (["-DWSEG_STAMPS=%d" % stamps] if stamps else [])
    if variant:      # its own object directory per (variant tag, stamps) so that two define sets never share objects
        objdir = os.path.join(HERE, "build", "variant_" + variant + ("_stamps%d" % stamps if stamps else ""))
        lib = os.path.join(LIBDIR, "libwseg_%s.so" % variant)
    if stamps or variant:
        flags = flags + list(defines)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "wseg.h"))
    srcs = sources()
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(s, o) or any(_newer(h, o) for h in headers) or not os.path.exists(o[:-2] + ".lint.json"):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        name = os.path.basename(o)[:-2]
        tmp = tempfile.mkdtemp(prefix="wseg_" + name + "_", dir=objdir)
        try:
            to = os.path.join(tmp, name + ".o")
            cmd = [HIPCC] + flags + ["-save-temps=obj", "-c", s, "-o", to]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
            digest = {"kernels": {}, "findings": [], "error": "no device assembly"}
            if asm:      # the lint lives under tools/ (not part of a deployment): without it the build still succeeds, the digest says why it is empty
                sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
                try:
                    import isa_lint
                    digest = isa_lint.analyze(os.path.join(tmp, asm[0]))
                except ImportError:
                    digest = {"kernels": {}, "findings": [], "error": "tools/isa_lint.py not found: assembly not linted"}
                finally:
                    sys.path.pop(0)
            with open(o[:-2] + ".lint.json", "w") as f:
                json.dump(digest, f)
            os.replace(to, o)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if jobs or not os.path.exists(lib):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    n = int(sys.argv[sys.argv.index("--stamps") + 1]) if "--stamps" in sys.argv else 0
    v = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else ""
    print(build(force="--force" in sys.argv, stamps=n, variant=v, defines=[x for x in sys.argv if x.startswith("-D")]))
