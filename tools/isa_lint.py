"""ISA / resource lint of the device code hipcc emits for libwseg (gfx950).

    python tools/isa_lint.py FILE.s [...]          # prints one line per kernel with scratch, spills or findings

`analyze(path)` reads the device assembly hipcc leaves behind with -save-temps (whisperseg_amd/build.py keeps a digest of it per
source file as build/<name>.lint.json) and returns

  kernels : {mangled name: {demangled, vgpr, agpr, sgpr, lds, scratch, vgpr_spill, sgpr_spill}}   from the amdhsa.kernels metadata
  findings: hazards around the inline-assembly MX MFMAs (`v_mfma_scale_f32_16x16x128_f8f6f4`, csrc/wseg_gemm.hip mfma_mx6_asm):
            hipcc treats an asm statement as one opaque instruction and pads none of its hazards (cdna_hip_programming.md §5.7):
              * a compiler-placed VALU write (v_mov, v_accvgpr_*, a conversion ...) of one of the instruction's A / B / scale registers
                within the 2 issue slots in front of `;;#ASMSTART`  (VALU write -> MFMA operand needs `s_nop 1`);
              * any compiler instruction other than an MFMA that reads or writes the accumulator tuple within 12 issue slots
                behind `;;#ASMEND` (MFMA D -> reader: 12 states on an 8-pass instruction; an accumulating MFMA taking D whole is 0).

tests/test_isa_lint.py asserts on the digest: no scratch and no spills in any kernel of the product's hot path (allow-listed
exceptions carry a measured reason), register counts under their occupancy steps, no findings."""
import json
import re
import subprocess
import sys

_BLOCK = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)")      # basic-block label or fall-through block comment
_FUNC = re.compile(r"^([A-Za-z_$][\w$.]*):")      # function label (hipcc appends "; @name")
_REG = re.compile(r"\b([vas])(?:(\d+)|\[(\d+):(\d+)\])")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        kind = m.group(1)
        if m.group(2) is not None:
            out.add((kind, int(m.group(2))))
        else:
            out.update((kind, i) for i in range(int(m.group(3)), int(m.group(4)) + 1))
    return out


def _split_operands(rest):
    """'v[0:3], v[4:9], v10 op_sel:[0,0]' -> ['v[0:3]', 'v[4:9]', 'v10 op_sel:[0,0]'] (commas inside brackets kept)."""
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


def _slots(mn, rest):
    if mn == "s_nop":
        try:
            return int(rest.strip(), 0) + 1
        except ValueError:
            return 1
    return 1


def _parse_metadata(lines, start):
    kernels, cur = {}, None
    keys = {".agpr_count": "agpr", ".vgpr_count": "vgpr", ".sgpr_count": "sgpr", ".group_segment_fixed_size": "lds",
            ".private_segment_fixed_size": "scratch", ".vgpr_spill_count": "vgpr_spill", ".sgpr_spill_count": "sgpr_spill",
            ".max_flat_workgroup_size": "wg"}
    for ln in lines[start:]:
        s = ln.strip()
        if s.startswith("- .agpr_count:") or (s.startswith("- .") and ln.startswith("  - ")):
            cur = {}
            s = s[2:]
        if cur is None:
            continue
        k, _, v = s.partition(":")
        k, v = k.strip(), v.strip()
        if k in keys and v:
            cur[keys[k]] = int(v)
        elif k == ".name":
            cur["name"] = v
        elif k == ".symbol" and "name" in cur:
            kernels[cur["name"]] = cur
    return kernels


def _demangle(names):
    if not names:
        return {}
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {n: re.sub(r"\((?:[^()]|\([^()]*\))*\)(?: \[clone[^\]]*\])?$", "", d) for n, d in zip(names, out)}


def analyze(path):
    with open(path, errors="replace") as f:
        lines = f.read().split("\n")
    meta_at = next((i for i, ln in enumerate(lines) if ln.startswith("amdhsa.kernels:")), len(lines))
    kernels = _parse_metadata(lines, meta_at)
    dem = _demangle(sorted(kernels))
    for n, k in kernels.items():
        k["demangled"] = dem.get(n, n)
    findings = []
    func, stream, in_asm = None, [], False      # stream: (in_asm, mnemonic, rest, line number)
    block_head = False
    in_loop, n_loop_scratch = False, 0          # per function: is the current basic block part of a loop (hipcc annotates every block
                                                # label with "in Loop: Header=..." / "=>This ... Loop Header"), scratch accesses in such blocks

    def flush_loops():
        # a scratch access inside a loop: a reload there waits (vmcnt retires in order) for every global load / LDS-DMA issued before
        # it — the r04 bring-up lost 4x to one such lane offset
        if func is not None and func in kernels:
            kernels[func]["scratch_in_loop"] = n_loop_scratch

    def flush():
        if func is None or not any(a and mn.startswith("v_mfma_scale") for a, mn, _, _ in stream):
            return
        for i, (a, mn, rest, no) in enumerate(stream):
            if not (a and mn.startswith("v_mfma_scale")):
                continue
            ops = _split_operands(rest)
            if len(ops) < 6:
                continue
            d, srcs = _regs(ops[0]), _regs(ops[1]) | _regs(ops[2]) | _regs(ops[4]) | _regs(ops[5].split(" ")[0])
            left, j = 2, i - 1                   # VALU write of an operand register right in front of the statement
            while j >= 0 and left > 0:
                a2, mn2, rest2, no2 = stream[j]
                if a2:
                    break                        # the previous asm MFMA: it writes accumulators only
                if mn2.startswith("v_") and not mn2.startswith("v_mfma"):
                    o2 = _split_operands(rest2)
                    if o2 and _regs(o2[0]) & srcs:
                        findings.append({"kernel": func, "line": no2, "kind": "valu write of an MX operand register %d slot(s) before the asm MFMA" % (3 - left),
                                         "text": (mn2 + " " + rest2).strip()})
                left -= _slots(mn2, rest2)
                j -= 1
            left, j = 12, i + 1                  # non-MFMA access to the accumulator tuple right behind it
            while j < len(stream) and left > 0:
                a2, mn2, rest2, no2 = stream[j]
                if not mn2.startswith("v_mfma") and _regs(rest2) & d and not mn2.startswith("s_"):
                    findings.append({"kernel": func, "line": no2, "kind": "non-MFMA access to the accumulator %d slot(s) behind the asm MFMA" % (13 - left),
                                     "text": (mn2 + " " + rest2).strip()})
                    break
                left -= _slots(mn2, rest2) * (4 if mn2.startswith("v_mfma") else 1)      # an MFMA occupies >= 4 states
                j += 1

    for no, ln in enumerate(lines[:meta_at], 1):
        fm = _FUNC.match(ln)
        if fm and not ln.startswith(".L"):
            flush()
            flush_loops()
            func, stream, in_asm = fm.group(1), [], False
            in_loop, n_loop_scratch = False, 0
            continue
        if _BLOCK.match(ln):                     # the annotation sits on the label line or on the comment lines right below it
            in_loop, block_head = "Loop" in ln, True
            continue
        if block_head:
            if ln.strip().startswith(";") and not ln.strip().startswith(";;#"):
                in_loop = in_loop or "Loop" in ln
                continue
            block_head = False
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s[0] in ".;" or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        mn, _, rest = s.partition(" ")
        stream.append((in_asm, mn, rest, no))
        if in_loop and (mn.startswith("scratch_") or (mn.startswith("buffer_") and "offen" in rest and "s[0:3]" in rest and " lds" not in rest)):
            n_loop_scratch += 1
    flush()
    flush_loops()
    for fnd in findings:
        fnd["kernel"] = dem.get(fnd["kernel"], fnd["kernel"])
    return {"kernels": kernels, "findings": findings}


def summarize(res, only_problems=True):
    rows = []
    for n, k in sorted(res["kernels"].items(), key=lambda kv: kv[1]["demangled"]):
        bad = k.get("scratch", 0) or k.get("vgpr_spill", 0) or k.get("sgpr_spill", 0)
        if bad or not only_problems:
            rows.append("scratch %4d (in loops: %d)  vspill %3d  sspill %3d  vgpr %3d  agpr %3d  sgpr %3d  lds %6d  %s" % (
                k.get("scratch", 0), k.get("scratch_in_loop", 0), k.get("vgpr_spill", 0), k.get("sgpr_spill", 0), k.get("vgpr", 0), k.get("agpr", 0),
                k.get("sgpr", 0), k.get("lds", 0), k["demangled"][:150]))
    for f in res["findings"]:
        rows.append("HAZARD %s: line %d: %s   [%s]" % (f["kind"], f["line"], f["text"], f["kernel"][:100]))
    return rows


if __name__ == "__main__":
    allk = "--all" in sys.argv
    for p in [a for a in sys.argv[1:] if not a.startswith("--")]:
        r = analyze(p)
        print("# %s: %d kernels" % (p, len(r["kernels"])))
        print("\n".join(summarize(r, not allk)))
        if "--json" in sys.argv:
            json.dump(r, open(p + ".lint.json", "w"))
