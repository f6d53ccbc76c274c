"""Host side of the split-precision modes ("bf16x3" / "f16x3"): operand-row layout and accuracy of the hi + lo split
(whisperseg_amd/engine.py::split_operand, mirrored on the device by csrc/wseg_common.h::op_st8 / x3_col)."""
import pytest
import torch

from whisperseg_amd.engine import DTYPES, SPLIT_BASE, is_gemm_weight, split_operand, to_engine_layout, unsplit_operand


@pytest.mark.parametrize("name", ["bf16x3", "f16x3"])
def test_split_layout_and_round_trip(name):
    base = SPLIT_BASE[name]
    g = torch.Generator().manual_seed(3)
    w = torch.randn(5, 96, generator=g) * 0.3
    s = split_operand(w, base)
    assert s.dtype == torch.int16 and s.shape == (5, 192)
    hi = w.to(base)
    lo = (w - hi.float()).to(base)
    words = s.view(base)
    for c in (0, 1, 31, 32, 33, 63, 64, 95):          # logical column c -> word (c // 32) * 64 + c % 32, its lo half 32 words later
        pos = (c // 32) * 64 + c % 32
        assert torch.equal(words[:, pos], hi[:, c]) and torch.equal(words[:, pos + 32], lo[:, c])
    back = unsplit_operand(s, base)
    assert torch.equal(back, hi.float() + lo.float())
    err = (back - w).abs()
    if name == "bf16x3":       # 8 + 8 mantissa bits and the fp32 exponent range: a uniform relative bound
        assert (err / w.abs().clamp_min(1e-30)).max().item() <= 2.0 ** -16
    else:                      # 11 + 11 bits, but lo halves below 2^-14 are f16 subnormals (spacing 2^-24): absolute floor
        assert (err <= 2.0 ** -22 * w.abs() + 2.0 ** -25).all()


def test_split_rejects_ragged_k():
    with pytest.raises(ValueError):
        split_operand(torch.zeros(4, 48), torch.bfloat16)


def test_engine_layout_splits_only_gemm_matrices():
    w = {"enc.0.qkv.w": torch.randn(384, 128), "enc.0.qkv.b": torch.randn(384), "dec.tok": torch.randn(256, 128),
         "dec.pos": torch.randn(448, 128), "enc.ln.g": torch.ones(128)}
    out = to_engine_layout(w, "bf16x3")
    assert {k for k, v in out.items() if v.dtype == torch.int16} == {"enc.0.qkv.w", "dec.tok"} == {k for k in w if is_gemm_weight(k)}
    assert out["enc.0.qkv.w"].shape == (384, 256) and out["dec.pos"].dtype == torch.float32
    plain = to_engine_layout(w, "f16")
    assert all(v.dtype == torch.float16 for v in plain.values())
    assert DTYPES["bf16x3"][0] == 3 and DTYPES["f16x3"][0] == 4


def test_half_split_saturates_instead_of_overflowing():
    """An operand beyond the fp16 range must not become inf - inf = NaN: it saturates at 65504 (what HF's own half path does to
    its hidden states); bfloat16 halves have the fp32 range and keep the value."""
    w = torch.tensor([[1.0e5, -3.0e6, 65504.0, 1.0] + [0.0] * 28])
    back = unsplit_operand(split_operand(w, torch.float16), torch.float16)
    assert torch.isfinite(back).all() and back[0, :4].tolist() == [65504.0, -65504.0, 65504.0, 1.0]
    back = unsplit_operand(split_operand(w, torch.bfloat16), torch.bfloat16)
    assert torch.allclose(back[0, :2], w[0, :2], rtol=2 ** -16)
