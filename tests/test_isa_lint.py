"""ISA / resource lint of libwseg's device code (CPU test: hipcc cross-compiles gfx950 here).

whisperseg_amd/build.py compiles every source with -save-temps and keeps tools/isa_lint.py's digest of the device assembly next
to the object the library is linked from (build/<name>.lint.json).  Asserted here, for EVERY kernel of the library (the default
mode f16m6 launches ~40 of them per decode step; a rule per family would rot):

  * no scratch and no VGPR spills — two kernels of the r04 hot path had them (the encoder attention of the split modes reloaded
    three spilled lane offsets in every key block behind an s_waitcnt vmcnt(0), draining its own K / V^T prefetch; the r04 bring-up
    of the mixed GEMM lost 4x to one such offset) — except the allow-listed entries below, each with a measured reason, and those
    must keep their scratch accesses OUT of loops;
  * register counts of the hot kernels under the occupancy step their launch geometry assumes;
  * no compiler-placed VALU write of an operand register in front of an inline-assembly MX MFMA and no non-MFMA access to its
    accumulator right behind it (hipcc pads no hazard of an asm statement: cdna_hip_programming.md §5.7).

The analysis itself is pinned on synthetic assembly with each defect planted."""
import glob
import json
import os
import re

import pytest

from conftest import ROOT

# kernel (regex on the demangled name) -> (max scratch bytes per lane, reason)
ALLOW_SCRATCH = {
    # (r04 needed one entry: dec_cross_attn_k24_kernel<*, 4> spilled a 64-bit pointer outside its loops; with its scores laid out
    # [position][beam] (one 16-byte LDS read per row) it needs 114 registers and no scratch)
}
# kernel regex -> VGPR ceiling (waves per SIMD the launch geometry counts on: MI355X_MICROARCH.md register-file table)
VGPR_CAPS = {
    r"gemm_h16_pp_kernel<": 256,                                   # 8 waves of 512 threads: 2 per SIMD
    r"gemm_w4_kernel<": 512,                                       # 4 waves: 1 per SIMD, accumulators in the AGPR half
    r"gemm_h16_persist_kernel<.*128, 128": 256,                    # 2 workgroups of 4 waves per CU: 2 waves per SIMD
    r"enc_attention_h16_kernel<.*true>": 168,                      # 3 workgroups per CU
    r"enc_attention_h16_kernel<.*false>": 128,                     # 4 workgroups per CU
    r"dec_self_attn_kernel<": 64,                                  # 8 single-wave workgroups per SIMD
    r"dec_cross_attn_bfp_kernel<": 128,                            # 4 workgroups per CU
    r"dec_cross_attn_k24_kernel<": 128,
    r"prompt_self_attn_kernel<": 64,
    r"dec_cross_attn_pk_kernel<": 128,
    r"layernorm_kernel<": 128,
}


@pytest.fixture(scope="module")
def digests():
    from whisperseg_amd import build as wbuild
    wbuild.build(verbose=False)                                    # no-op when objects and digests are current
    out = {}
    for f in sorted(glob.glob(os.path.join(ROOT, "whisperseg_amd", "build", "*.lint.json"))):
        with open(f) as fh:
            out[os.path.basename(f)] = json.load(fh)
    assert {"wseg_gemm.lint.json", "wseg_enc.lint.json", "wseg_dec.lint.json", "wseg_logmel.lint.json"} <= set(out)
    return out


def all_kernels(digests):
    for f, d in digests.items():
        assert not d.get("error"), (f, d.get("error"))
        for k in d["kernels"].values():
            yield f, k


def test_every_kernel_was_digested(digests):
    ks = list(all_kernels(digests))
    assert len(ks) > 300
    names = " ".join(k["demangled"] for _, k in ks)
    for must in ("gemm_h16_pp_kernel<wseg::M6", "enc_attention_h16_kernel<wseg::f16_t, wseg::M6, true>", "dec_cross_attn_bfp_kernel<wseg::M6, 4>",
                 "dec_self_attn_kernel<float, wseg::M6>", "logmel_fft_kernel", "splitk_reduce_resid_ln_kernel<wseg::M6>", "beam_step_kernel"):
        assert must in names, must
    for _, k in ks:
        assert "scratch_in_loop" in k and k["vgpr"] > 0, k["demangled"]


def test_no_scratch_and_no_spills(digests):
    bad = []
    for f, k in all_kernels(digests):
        if not (k["scratch"] or k["vgpr_spill"]):
            continue
        allow = [v for pat, v in ALLOW_SCRATCH.items() if re.search(pat, k["demangled"])]
        if allow and k["scratch"] <= allow[0][0] and k["scratch_in_loop"] == 0:
            continue
        bad.append((k["demangled"], k["scratch"], k["vgpr_spill"], k["scratch_in_loop"]))
    assert not bad, bad


def test_register_counts_under_their_occupancy_steps(digests):
    seen = set()
    for _, k in all_kernels(digests):
        for pat, cap in VGPR_CAPS.items():
            if re.search(pat, k["demangled"]):
                seen.add(pat)
                assert k["vgpr"] <= cap, (k["demangled"], k["vgpr"], k["agpr"], cap)      # (.vgpr_count is the unified total: it includes the AGPRs)
    assert len(seen) >= len(VGPR_CAPS) - 1, sorted(set(VGPR_CAPS) - seen)      # (gemm_w4_kernel may not exist in every build)


def test_no_hazard_around_inline_asm_mfmas(digests):
    finds = [(f, x) for f, d in digests.items() for x in d["findings"]]
    assert not finds, finds[:5]


SYNTH = """
\t.text
_Z4goodv:                               ; @_Z4goodv
\tv_mov_b32_e32 v20, v1
\ts_nop 1
\t;;#ASMSTART
\tv_mfma_scale_f32_16x16x128_f8f6f4 v[0:3], v[10:15], v[20:25], v[0:3], v30, v31 op_sel_hi:[0,0,0] cbsz:2 blgp:2
\t;;#ASMEND
\tv_mfma_f32_16x16x32_f16 v[0:3], v[40:43], v[44:47], v[0:3]
\ts_endpgm
_Z4bad1v:                               ; @_Z4bad1v
\tv_mov_b32_e32 v20, v1
\t;;#ASMSTART
\tv_mfma_scale_f32_16x16x128_f8f6f4 v[0:3], v[10:15], v[20:25], v[0:3], v30, v31 op_sel_hi:[0,0,0] cbsz:2 blgp:2
\t;;#ASMEND
\ts_nop 7
\ts_nop 7
\tv_add_f32_e32 v50, v0, v0
\ts_endpgm
_Z4bad2v:                               ; @_Z4bad2v
\t;;#ASMSTART
\tv_mfma_scale_f32_16x16x128_f8f6f4 v[0:3], v[10:15], v[20:25], v[0:3], v30, v31 op_sel_hi:[0,0,0] cbsz:2 blgp:2
\t;;#ASMEND
\ts_nop 3
\tv_add_f32_e32 v50, v1, v1
\ts_endpgm
_Z4loopv:                               ; @_Z4loopv
\tscratch_store_dword off, v7, off        ; 4-byte Folded Spill
.LBB3_1:                                ; %.loopexit
                                        ; =>This Inner Loop Header: Depth=1
\tscratch_load_dword v53, off, off        ; 4-byte Folded Reload
\ts_cbranch_scc1 .LBB3_1
.LBB3_2:                                ; %.exit
\tscratch_load_dword v54, off, off        ; 4-byte Folded Reload
\ts_endpgm
amdhsa.kernels:
  - .agpr_count:     0
    .name:           _Z4loopv
    .private_segment_fixed_size: 4
    .sgpr_count:     10
    .symbol:         _Z4loopv.kd
    .vgpr_count:     54
    .vgpr_spill_count: 1
"""


def test_the_lint_finds_planted_defects(tmp_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import isa_lint
    finally:
        sys.path.pop(0)
    p = tmp_path / "synth.s"
    p.write_text(SYNTH)
    r = isa_lint.analyze(str(p))
    kinds = {(f["kernel"], f["kind"].split(" ")[0]) for f in r["findings"]}
    assert kinds == {("_Z4bad1v", "valu"), ("_Z4bad2v", "non-MFMA")}, r["findings"]      # (names demangle only for kernels with metadata)
    k = r["kernels"]["_Z4loopv"]
    assert k["scratch"] == 4 and k["vgpr_spill"] == 1 and k["scratch_in_loop"] == 1
