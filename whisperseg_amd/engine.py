"""Host-side driver of the libwseg model: weight preparation, workspace management, generate().

PyTorch-ROCm is plumbing here (device memory, streams); every FLOP of the model runs in libwseg's
hand-written HIP kernels through the C-ABI of include/wseg.h.  There is no CPU fallback.

Replaces what the reference obtains from `WhisperForConditionalGeneration.from_pretrained(...)` and
`model.generate(...)` (reference model.py:633-636, 655-666).
"""
import ctypes as C
import json
import os

import torch

from . import _lib

# name -> (wseg_dtype, torch dtype of parameters and activations).  "bf16x3" / "f16x3" are the split-precision modes
# (include/wseg.h): fp32 everywhere except that GEMM operands travel as hi + lo 16-bit pairs and are multiplied with three
# MFMAs per product; their weight MATRICES are attached pre-split (split_operand), everything else is float32.
# "f16m6" is the MIXED split-precision mode: f16x3 everywhere outside the GEMMs; a GEMM takes hi*hi on the IEEE-half matrix cores and the
# two cross terms on the block-scaled MX matrix cores (fp6 operands, 4x the 16-bit rate): its weight matrices are built as f16x3
# operand rows and then converted ON THE DEVICE to "M6 rows" (wseg_convert_operand; same byte size), the token embedding is
# attached a second time as plain fp32 ("dec.tok.f32") for the embedding lookup.
DTYPES = {"f32": (0, torch.float32), "bf16": (1, torch.bfloat16), "f16": (2, torch.float16),
          "bf16x3": (3, torch.float32), "f16x3": (4, torch.float32), "f16m6": (5, torch.float32)}
SPLIT_BASE = {"bf16x3": torch.bfloat16, "f16x3": torch.float16, "f16m6": torch.float16}
MX_MODES = ("f16m6",)


def is_gemm_weight(name):
    """Tensors that are the W operand of a GEMM (the token embedding doubles as the LM head)."""
    return name.endswith(".w") or name == "dec.tok"


def split_operand(w, base):
    """fp32 [N, K] (K % 32 == 0) -> int16 [N, 2K] operand rows of the split-precision modes: hi = rn(w), lo = rn(w - hi) in
    the 16-bit type `base`, every 32 logical columns stored as [32 hi | 32 lo] (csrc/wseg_common.h)."""
    n, k = w.shape
    if k % 32:
        raise ValueError("split_operand: K must be a multiple of 32")
    w = w.float()
    if base == torch.float16:      # saturate instead of inf - inf = NaN (as the device-side producers do)
        w = w.clamp(-65504.0, 65504.0)
    hi = w.to(base)
    lo = (w - hi.float()).to(base)
    out = torch.stack([hi.view(n, k // 32, 32), lo.view(n, k // 32, 32)], dim=2)
    return out.reshape(n, 2 * k).view(torch.int16).contiguous()


def unsplit_operand(t, base):
    """Inverse of split_operand (hi + lo is exact in fp32): int16 [N, 2K] -> fp32 [N, K]."""
    n, k2 = t.shape
    v = t.view(base).view(n, k2 // 64, 2, 32).float()
    return (v[:, :, 0] + v[:, :, 1]).reshape(n, k2 // 2)


def unsplit_m6(t, weight_order=False):
    """M6 rows (csrc/wseg_common.h; int16 [N, 2K] as produced by wseg_convert_operand or written by a kernel) -> fp32 [N, K]:
    hi half + the fp6 (e2m3, block-scaled) image of the lo half — what a GEMM of the mixed mode sees of the operand, to ~2^-15
    relative.  Test helper."""
    n, k2 = t.shape
    k = k2 // 2
    raw = t.contiguous().view(torch.uint8).view(n, k // 64, 256)
    hi = raw[:, :, :128].contiguous().view(torch.float16).float()                      # [n, k/64, 64]
    mx = raw[:, :, 128:].to(torch.int64)                                                # [n, k/64, 128] bytes
    lo = torch.zeros(n, k // 64, 64, device=t.device)
    for chunk in range(2):
        g = (2 if weight_order else 0) + chunk
        cb = mx[:, :, 32 * g:32 * g + 24]                                               # 24 code bytes = 32 codes, little-endian 6-bit
        word = sum(cb[:, :, i::3] << (8 * i) for i in range(3))                          # 8 x 24-bit words holding 4 codes each
        codes = torch.stack([(word >> (6 * j)) & 63 for j in range(4)], dim=-1).reshape(n, k // 64, 32)
        sign = 1.0 - 2.0 * ((codes >> 5) & 1).float()
        e, mnt = (codes >> 3) & 3, (codes & 7).float()
        val = torch.where(e == 0, mnt / 8.0, (1.0 + mnt / 8.0) * torch.exp2((e - 1).float()))
        scale = torch.exp2(mx[:, :, 32 * g + 24].float() - 127.0).unsqueeze(-1)
        lo[:, :, 32 * chunk:32 * chunk + 32] = sign * val * scale
    return (hi + lo).reshape(n, k)


def to_engine_layout(weights, dtype):
    """{name: fp32 tensor in libwseg layout} -> tensors as the engine of mode `dtype` attaches them."""
    base = SPLIT_BASE.get(dtype)
    td = DTYPES[dtype][1]
    if base is None:
        return {k: v.to(td).contiguous() for k, v in weights.items()}
    return {k: (split_operand(v, base) if is_gemm_weight(k) else v.float().contiguous()) for k, v in weights.items()}

# generation_config.suppress_tokens of the multilingual Whisper checkpoints WhisperSeg fine-tunes
# (saved with the trained models: reference docs/WhisperSeg_Training_Pipeline.ipynb, generation params).
DEFAULT_SUPPRESS_TOKENS = [
    1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 359, 503, 522, 542, 873,
    893, 902, 918, 922, 931, 1350, 1853, 1982, 2460, 2627, 3246, 3253, 3268, 3536, 3846, 3961, 4183, 4667, 6585, 6647,
    7273, 9061, 9383, 10428, 10929, 11938, 12033, 12331, 12562, 13793, 14157, 14635, 15265, 15618, 16553, 16604, 18362,
    18956, 20075, 21675, 22520, 26130, 26161, 26435, 28279, 29464, 31650, 32302, 32470, 36865, 42863, 47425, 49870,
    50254, 50258, 50358, 50359, 50360, 50361, 50362]
DEFAULT_BEGIN_SUPPRESS_TOKENS = [220, 50257]
# Window slots decoded concurrently (in-flight batching; include/wseg.h).  Throughput grows with the concurrency until
# ~1024 windows (whisperseg-large, 4 beams: 14.2 / 15.9 / 17.0 k audio-sec/s at 256 / 512 / 1024 slots — more GEMM rows per
# weight pass, less launch tail in the attention streams).  Since ABI 5 the self-attention K/V are paged: the workspace holds
# min(max_length, 64) positions per slot on average whatever max_length is, so segment()'s default max_length = 448 keeps the
# same 1024 slots as a 35-position decode (whisperseg-large, 4 beams, split-precision mode: ~123 MiB of cross K/V + 1.3 MiB per
# pooled position per slot).
DEFAULT_SLOTS = 1024
# consecutive small calls before a much larger workspace is given back (a folder loop over files of very different lengths, or
# encode() between generate() calls, must not free and re-allocate tens of GB — and re-capture the step graph — on every flip)
SHRINK_AFTER_CALLS = 4


def _round_up(v, a):
    return (v + a - 1) // a * a


def geometry_from_config(cfg):
    """cfg: dict in HF config.json vocabulary -> libwseg geometry dict."""
    d = cfg["d_model"]
    heads = cfg["encoder_attention_heads"]
    if cfg.get("decoder_attention_heads", heads) != heads or cfg.get("decoder_ffn_dim", cfg["encoder_ffn_dim"]) != cfg["encoder_ffn_dim"]:
        raise ValueError("encoder/decoder head count and ffn width must match")
    spec_cols = cfg.get("total_spec_columns", 2 * cfg.get("max_source_positions", 500))
    return dict(d_model=d, n_heads=heads, enc_layers=cfg["encoder_layers"], dec_layers=cfg["decoder_layers"],
                ffn=cfg["encoder_ffn_dim"], vocab=cfg["vocab_size"], n_mels=cfg.get("num_mel_bins", 80),
                spec_cols=spec_cols, enc_positions=cfg.get("max_source_positions", spec_cols // 2),
                dec_positions=cfg.get("max_target_positions", 448))


def prepare_weights(sd, geo, torch_dtype, device, split_base=None):
    """HF-named state dict -> {libwseg tensor name: contiguous device tensor} (layouts of include/wseg.h).
    split_base (torch.bfloat16 / torch.float16): split-precision mode — GEMM weight matrices become operand rows.

    qkv.w  [3d, d]   rows = q | k | v projections;   qkv.b has zeros for k (k_proj has no bias)
    ckv.w  [2d, d]   rows = cross-attention k | v
    conv1.w [d, Kp]  k = tap * n_mels + channel, zero padded to a multiple of 64
    conv2.w [d, 3d]  k = tap * d + channel
    dec.tok [Vp, d]  token embedding == tied LM head, rows padded to a multiple of 128
    """
    d, n_mels = geo["d_model"], geo["n_mels"]
    dev = torch.device(device)

    def g(name):      # one tensor at a time: `sd` may be a LazyCheckpoint reading shards on demand
        return sd[name].to(device=dev, dtype=torch.float32)

    class _Out(dict):     # every prepared tensor is cast to the model dtype as soon as it exists (no full fp32 copy)
        def __setitem__(self, key, value):
            if split_base is not None and is_gemm_weight(key):
                dict.__setitem__(self, key, split_operand(value, split_base))
            else:
                dict.__setitem__(self, key, value.to(torch_dtype).contiguous())

    out = _Out()
    kp1 = _round_up(3 * n_mels, 64)
    w1 = g("model.encoder.conv1.weight").permute(0, 2, 1).reshape(d, 3 * n_mels)
    out["enc.conv1.w"] = torch.nn.functional.pad(w1, (0, kp1 - 3 * n_mels))
    out["enc.conv1.b"] = g("model.encoder.conv1.bias")
    out["enc.conv2.w"] = g("model.encoder.conv2.weight").permute(0, 2, 1).reshape(d, 3 * d)
    out["enc.conv2.b"] = g("model.encoder.conv2.bias")
    out["enc.pos"] = g("model.encoder.embed_positions.weight")[: geo["enc_positions"]]
    out["enc.ln.g"] = g("model.encoder.layer_norm.weight")
    out["enc.ln.b"] = g("model.encoder.layer_norm.bias")
    zeros = torch.zeros(d, device=dev)

    def attn(dst, src, fused_kv_only=False):
        q_w, k_w, v_w = g(src + "q_proj.weight"), g(src + "k_proj.weight"), g(src + "v_proj.weight")
        q_b, v_b = g(src + "q_proj.bias"), g(src + "v_proj.bias")
        if fused_kv_only:
            out[dst + "cq.w"], out[dst + "cq.b"] = q_w, q_b
            out[dst + "ckv.w"] = torch.cat([k_w, v_w], 0)
            out[dst + "ckv.b"] = torch.cat([zeros, v_b], 0)
            out[dst + "co.w"], out[dst + "co.b"] = g(src + "out_proj.weight"), g(src + "out_proj.bias")
        else:
            out[dst + "qkv.w"] = torch.cat([q_w, k_w, v_w], 0)
            out[dst + "qkv.b"] = torch.cat([q_b, zeros, v_b], 0)
            out[dst + "o.w"], out[dst + "o.b"] = g(src + "out_proj.weight"), g(src + "out_proj.bias")

    def ln(dst, src):
        out[dst + ".g"], out[dst + ".b"] = g(src + ".weight"), g(src + ".bias")

    def mlp(dst, src):
        for n in ("fc1", "fc2"):
            out[f"{dst}{n}.w"], out[f"{dst}{n}.b"] = g(f"{src}{n}.weight"), g(f"{src}{n}.bias")

    for i in range(geo["enc_layers"]):
        s, t = f"model.encoder.layers.{i}.", f"enc.{i}."
        ln(t + "ln1", s + "self_attn_layer_norm")
        attn(t, s + "self_attn.")
        ln(t + "ln2", s + "final_layer_norm")
        mlp(t, s)
    vp = _round_up(geo["vocab"], 128)
    tok = g("model.decoder.embed_tokens.weight")
    out["dec.tok"] = torch.nn.functional.pad(tok, (0, 0, 0, vp - tok.shape[0]))
    out["dec.pos"] = g("model.decoder.embed_positions.weight")[: geo["dec_positions"]]
    ln("dec.ln", "model.decoder.layer_norm")
    for i in range(geo["dec_layers"]):
        s, t = f"model.decoder.layers.{i}.", f"dec.{i}."
        ln(t + "ln1", s + "self_attn_layer_norm")
        attn(t, s + "self_attn.")
        ln(t + "ln2", s + "encoder_attn_layer_norm")
        attn(t, s + "encoder_attn.", fused_kv_only=True)
        ln(t + "ln3", s + "final_layer_norm")
        mlp(t, s)
    return dict(out)


def random_weights(geo, torch_dtype, device, seed=0, split_base=None):
    """Seeded random weights generated directly on the device in libwseg layout (benchmarks:
    no checkpoint exists offline).  Same distributions as oracle.whisper_ref.random_state_dict.
    split_base: split-precision mode (torch_dtype float32) — the fp32 draws of the GEMM matrices become operand rows, so an
    "f32" engine with the same seed holds exactly the values these hi + lo pairs approximate."""
    gen = torch.Generator(device=device).manual_seed(seed)
    d, f, nm = geo["d_model"], geo["ffn"], geo["n_mels"]
    vp, kp1 = _round_up(geo["vocab"], 128), _round_up(3 * nm, 64)

    def rn(*shape, s=0.02):
        return (torch.randn(*shape, generator=gen, device=device, dtype=torch.float32) * s).to(torch_dtype)

    out = {"enc.conv1.w": rn(d, kp1, s=0.05), "enc.conv1.b": rn(d), "enc.conv2.w": rn(d, 3 * d), "enc.conv2.b": rn(d),
           "enc.pos": rn(geo["enc_positions"], d), "dec.tok": rn(vp, d, s=0.05), "dec.pos": rn(geo["dec_positions"], d)}
    out["enc.conv1.w"][:, 3 * nm:] = 0

    def ln(p):
        out[p + ".g"] = (1.0 + torch.randn(d, generator=gen, device=device) * 0.1).to(torch_dtype)
        out[p + ".b"] = rn(d, s=0.1)

    def lin(p, n, k):
        out[p + ".w"] = rn(n, k, s=k ** -0.5)
        out[p + ".b"] = rn(n)

    ln("enc.ln"); ln("dec.ln")
    for side, n in (("enc", geo["enc_layers"]), ("dec", geo["dec_layers"])):
        for i in range(n):
            p = f"{side}.{i}."
            ln(p + "ln1"); lin(p + "qkv", 3 * d, d); lin(p + "o", d, d); ln(p + "ln2")
            out[p + "qkv.b"][d:2 * d] = 0
            if side == "dec":
                lin(p + "cq", d, d); lin(p + "ckv", 2 * d, d); lin(p + "co", d, d); ln(p + "ln3")
                out[p + "ckv.b"][:d] = 0
            lin(p + "fc1", f, d); lin(p + "fc2", d, f)
    if split_base is not None:
        for k in list(out):
            if is_gemm_weight(k):
                out[k] = split_operand(out[k], split_base)
    return out


class Engine:
    """One model replica on one GPU."""

    def __init__(self, geo, weights, device="cuda:0", dtype="bf16"):
        self.lib = _lib.load(require_device=True)
        self.device = torch.device(device)
        self.geo = dict(geo)
        self.dtype_name = dtype
        self.dtype_id, self.torch_dtype = DTYPES[dtype]
        cfg = _lib.ModelConfig(**self.geo, dtype=self.dtype_id)
        handle = C.c_void_p()
        _lib.check(self.lib.wseg_model_create(C.byref(cfg), C.byref(handle)))
        self.handle = handle
        self.split_base = SPLIT_BASE.get(dtype)
        self.mx = dtype in MX_MODES
        self._x3_host = None
        if self.mx:
            weights = self._to_m6(weights)
        self.weights = weights          # keep the device tensors alive
        for name, t in weights.items():
            want = torch.int16 if (self.split_base is not None and is_gemm_weight(name)) else self.torch_dtype
            if name == "dec.tok.f32":
                want = torch.float32
            if t.device != self.device or t.dtype != want or not t.is_contiguous():
                raise ValueError(f"weight {name}: wrong device/dtype/layout")
            _lib.check(self.lib.wseg_model_set_tensor(handle, name.encode(), t.data_ptr(), t.numel() * t.element_size()))
        _lib.check(self.lib.wseg_model_ready(handle))
        self._ws = None
        self._small_calls = 0

    def _to_m6(self, weights):
        """f16x3 operand rows of the GEMM weight matrices -> M6 rows (device-side conversion, weight order); the hi | lo rows move
        to host memory (weights_f32 / sibling rebuild the exact values from them); + the fp32 embedding table."""
        out, self._x3_host = {}, {}
        with torch.cuda.device(self.device):
            for name, t in weights.items():
                if not is_gemm_weight(name):
                    out[name] = t
                    continue
                if t.dtype != torch.int16 or t.device != self.device or not t.is_contiguous():
                    raise ValueError(f"weight {name}: wrong device/dtype/layout")
                n, k2 = t.shape
                dst = torch.empty_like(t)
                _lib.check(self.lib.wseg_convert_operand(t.data_ptr(), dst.data_ptr(), n, k2 // 2, 1, _lib.stream_ptr()))
                out[name] = dst
                if name == "dec.tok":
                    out["dec.tok.f32"] = unsplit_operand(t, self.split_base).contiguous()
                self._x3_host[name] = t.cpu()
            torch.cuda.synchronize(self.device)
        return out

    @classmethod
    def from_state_dict(cls, sd, hf_config, device="cuda:0", dtype="bf16"):
        geo = geometry_from_config(hf_config)
        return cls(geo, prepare_weights(sd, geo, DTYPES[dtype][1], device, SPLIT_BASE.get(dtype)), device, dtype)

    @classmethod
    def from_pretrained(cls, model_dir, device="cuda:0", dtype="bf16"):
        """Read an HF-style checkpoint directory: config.json + model.safetensors | pytorch_model.bin, single-file or
        sharded with an index (whisperseg_amd/checkpoint.py; what the reference's save_pretrained writes, model.py:59-74)."""
        from .checkpoint import LazyCheckpoint
        with open(os.path.join(model_dir, "config.json")) as f:
            hf_config = json.load(f)
        ckpt = LazyCheckpoint(model_dir)
        try:
            return cls.from_state_dict(ckpt, hf_config, device, dtype)
        finally:
            ckpt.close()

    @classmethod
    def random(cls, hf_config, device="cuda:0", dtype="bf16", seed=0):
        geo = geometry_from_config(hf_config)
        return cls(geo, random_weights(geo, DTYPES[dtype][1], device, seed, SPLIT_BASE.get(dtype)), device, dtype)

    def weights_f32(self):
        """{name: fp32 tensor in libwseg layout} holding exactly the values this engine computes with (operand rows of the
        split-precision modes are read back as hi + lo)."""
        out = {}
        for k, v in self.weights.items():
            if k == "dec.tok.f32":
                continue
            if self._x3_host is not None and k in self._x3_host:      # M6 rows are lossy in the lo part: the hi | lo rows are kept on the host
                v = self._x3_host[k].to(self.device)
            out[k] = unsplit_operand(v, self.split_base) if v.dtype == torch.int16 else v.float()
        return out

    def sibling(self, dtype):
        """An engine of another mode over the SAME weight values (rounded to that mode's storage type)."""
        return Engine(self.geo, to_engine_layout(self.weights_f32(), dtype), self.device, dtype)

    def exact_reference(self):
        """The exact-parity f32 engine over this engine's weight values (bench.py's self-check, tools/logit_error.py)."""
        return self.sibling("f32")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.wseg_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def _workspace(self, n_slots, num_beams, max_length, kv_positions=0, may_shrink=True):
        need = self.lib.wseg_workspace_bytes_kv(self.handle, n_slots, num_beams, max_length, int(kv_positions or 0))
        if need == 0:
            raise _lib.WsegError("wseg_workspace_bytes rejected the request")
        held = 0 if self._ws is None else self._ws.numel()
        # grow when too small.  Give a much larger one back only after SHRINK_AFTER_CALLS consecutive generate() calls that needed
        # less than a quarter of it (a single long file must not pin ~100 GB for every later short call, but alternating large
        # and small requests must not thrash either); encode() never shrinks the decode workspace (may_shrink=False).
        small = held > 4 * need and held > (2 << 30)
        if may_shrink:
            self._small_calls = self._small_calls + 1 if small else 0
        if held < need or (small and may_shrink and self._small_calls >= SHRINK_AFTER_CALLS):
            self._ws = None
            self._small_calls = 0
            if held >= (1 << 30):
                # hand the old block back to the driver first: kept in torch's cache it could not be reused for the larger request
                torch.cuda.empty_cache()
            try:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            except torch.OutOfMemoryError as exc:
                raise _lib.WsegOutOfMemory(
                    f"cannot allocate the {need / 2 ** 30:.1f}-GiB decode workspace for {n_slots} window slots on {self.device} (the slot "
                    "count is derived from the device's TOTAL memory so that results never depend on what else is resident; "
                    "lower it with segmenter.max_slots / $WSEG_SLOTS when the GPU is shared)") from exc
        return self._ws

    def release_workspace(self):
        """Free the decode workspace (it is re-allocated by the next call)."""
        self._ws = None
        self._small_calls = 0

    def weight_bytes(self):
        return sum(t.numel() * t.element_size() for t in self.weights.values())

    def pick_slots(self, n_windows, num_beams, max_length, n_slots=None, kv_positions=0):
        """Window slots for a generate call: min(n_windows, cap) where cap is `n_slots`, else $WSEG_SLOTS, else DEFAULT_SLOTS,
        halved until the workspace fits in 80 % of the device's TOTAL memory minus this engine's weights.  Deliberately NOT a
        function of the memory that happens to be free: in the 16-bit modes the GEMM plans (and with them the last bits of the
        logits) follow the row count = slots x beams, so the slot count must be a pure function of (geometry, mode, call
        parameters, device) for a recording to decode the same way run after run; the count used is
        reported by last_stats()["n_slots"].  When the GPU is shared and the allocation fails, _workspace raises with the
        remedy.  The reference bounds memory with `batch_size`; here that role is played by `n_slots` / $WSEG_SLOTS
        (SegmenterBase.max_slots)."""
        cap = int(n_slots or os.environ.get("WSEG_SLOTS", 0) or DEFAULT_SLOTS)
        s = max(1, min(int(n_windows), cap))
        total = torch.cuda.get_device_properties(self.device).total_memory
        budget = 0.8 * total - self.weight_bytes()
        while s > 1 and self.lib.wseg_workspace_bytes_kv(self.handle, s, num_beams, max_length, int(kv_positions or 0)) > budget:
            s = (s + 1) // 2
        return s

    def encode(self, feats):
        """feats float32 device tensor [W, 80, 1000] -> [W, 500, d] in the model dtype (float32 in the split-precision modes)."""
        feats = feats.to(device=self.device, dtype=torch.float32).contiguous()
        W = feats.shape[0]
        ws = self._workspace(W, 1, 8, may_shrink=False)
        out = torch.empty((W, self.geo["enc_positions"], self.geo["d_model"]), dtype=self.torch_dtype, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.wseg_encode(self.handle, feats.data_ptr(), W, ws.data_ptr(), ws.numel(), out.data_ptr(),
                                            _lib.stream_ptr()))
        return out

    def generate(self, feats, prompt, eos_token_id, pad_token_id, max_length=448, num_beams=4, length_penalty=1.0,
                 suppress_tokens=(), begin_suppress_tokens=(), return_first_logits=False, n_slots=None, refill_min=0,
                 lookahead=0, window_max_length=None, encoder_output=None, top_k=1, top_p=1.0, seed=0, kv_positions=0):
        """Greedy / beam-search decode of ALL windows of `feats` [N, 80, 1000] through `n_slots` window slots with
        in-flight refill (a finished window's slot goes to the next queued window; wseg_generate; see pick_slots for the
        default slot count).  `top_k` in 2..16 with
        num_beams == 1 samples (top-k, then nucleus `top_p`) with a counter-based generator keyed by `seed`.
        `kv_positions`: average self-attention K/V positions per slot to provision in the paged pool (0: min(max_length, 64);
        >= max_length: every slot can reach max_length at once); when the pool runs short the engine preempts and re-decodes
        the youngest window (last_stats()["n_preemptions"]).
        Returns (tokens int32 [N, max_length] on device, lengths int32 [N])."""
        feats = feats.to(device=self.device, dtype=torch.float32).contiguous()
        W = feats.shape[0]
        max_length = int(min(max_length, self.geo["dec_positions"]))
        with torch.cuda.device(self.device):
            slots = self.pick_slots(W, num_beams, max_length, n_slots, kv_positions)
            pinned = n_slots is not None or bool(os.environ.get("WSEG_SLOTS"))
            self.slots_halved = 0
            while True:
                try:
                    ws = self._workspace(slots, num_beams, max_length, kv_positions)
                    break
                except _lib.WsegOutOfMemory:
                    # The budget ignores other residents of the GPU (a sibling engine, another process).  Only the allocation failure
                    # itself is answered with fewer slots (any other error stands), never when the caller pinned the count
                    # (n_slots / segmenter.max_slots / $WSEG_SLOTS), and only in the exact mode — whose tokens are structurally
                    # independent of the slot count — and the split modes, where that independence is TESTED on the committed
                    # geometries (trained models: exact; flat random-weight logits: 4 094 of 4 096 windows, bench `check`), so a
                    # near-tie may resolve differently: the warning says so and last_stats() records it (n_slots, slots_halved).
                    # The plain 16-bit modes' logits follow the row count visibly: there the error stands.
                    if slots <= 1 or pinned or self.dtype_name in ("bf16", "f16"):
                        raise
                    import warnings
                    warnings.warn(f"decode workspace for {slots} window slots does not fit beside the GPU's other allocations: "
                                  f"retrying with {(slots + 1) // 2} slots (mode {self.dtype_name}: the GEMM plans follow the slot count; "
                                  + ("tokens are unaffected" if self.dtype_name == "f32" else
                                     "tokens are tested equal on the committed geometries, a near-tie between two hypotheses may still resolve differently")
                                  + "; pin the count with max_slots / $WSEG_SLOTS for run-to-run identical plans)")
                    slots = (slots + 1) // 2
                    self.slots_halved += 1
        sup = torch.tensor(list(suppress_tokens) or [0], dtype=torch.int32, device=self.device)
        bsup = torch.tensor(list(begin_suppress_tokens) or [0], dtype=torch.int32, device=self.device)
        gp = _lib.GenerateParams()
        for i, t in enumerate(prompt):
            gp.prompt[i] = int(t)
        gp.prompt_len = len(prompt)
        gp.eos_token_id, gp.pad_token_id = int(eos_token_id), int(pad_token_id)
        gp.max_length, gp.num_beams, gp.length_penalty = max_length, int(num_beams), float(length_penalty)
        gp.suppress_tokens, gp.n_suppress = sup.data_ptr(), len(suppress_tokens)
        gp.begin_suppress_tokens, gp.n_begin_suppress = bsup.data_ptr(), len(begin_suppress_tokens)
        gp.n_slots, gp.refill_min, gp.lookahead = int(slots), int(refill_min), int(lookahead)
        gp.top_k, gp.top_p, gp.seed = int(top_k), float(top_p), int(seed) & (2 ** 64 - 1)      # sampling: num_beams == 1 and top_k > 1
        wml = None
        if window_max_length is not None:      # per-window total-length caps (int32 [W])
            wml = torch.as_tensor(window_max_length, dtype=torch.int32).to(self.device).contiguous()
            if wml.numel() != W:
                raise ValueError("window_max_length needs one entry per window")
            gp.window_max_length = wml.data_ptr()
        enc = None
        if encoder_output is not None:         # precomputed encoder states [W, 500, d] (padded to the GEMM's 256-row granularity)
            enc = encoder_output.to(device=self.device, dtype=self.torch_dtype).reshape(W * self.geo["enc_positions"], self.geo["d_model"])
            pad = _round_up(enc.shape[0], 256) - enc.shape[0]
            enc = torch.nn.functional.pad(enc, (0, 0, 0, pad)).contiguous()
            gp.encoder_output = enc.data_ptr()
        tokens = torch.empty((W, max_length), dtype=torch.int32, device=self.device)
        lengths = torch.empty((W,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.wseg_generate(self.handle, feats.data_ptr(), W, C.byref(gp), ws.data_ptr(), ws.numel(),
                                              tokens.data_ptr(), lengths.data_ptr(), _lib.stream_ptr()))
            if return_first_logits:
                if W > slots:
                    raise ValueError("return_first_logits needs every window to start together (n_slots >= windows)")
                fl = torch.empty((W * num_beams, self.geo["vocab"]), dtype=torch.float32, device=self.device)
                _lib.check(self.lib.wseg_debug_first_logits(self.handle, ws.data_ptr(), fl.data_ptr(), W * num_beams,
                                                            _lib.stream_ptr()))
                return tokens, lengths, fl
        return tokens, lengths

    def last_stats(self):
        """Scheduler statistics of the last generate call: dict(n_windows, n_slots, n_steps, n_admissions,
        slot_steps_active, slot_steps_total, occupancy, steady_occupancy = occupancy while windows were still queued,
        kv_units_total / kv_units_peak = pool units of the paged self-attention K/V (one unit = an 8-position page of every beam
        of a slot), n_preemptions)."""
        st = _lib.GenerateStats()
        _lib.check(self.lib.wseg_last_stats(self.handle, C.byref(st)))
        out = {k: int(getattr(st, k)) for k, _ in st._fields_ if not k.endswith("_")}
        out["slots_halved"] = int(getattr(self, "slots_halved", 0))      # times generate() halved the slot count after an allocation failure
        out["occupancy"] = out["slot_steps_active"] / max(1, out["slot_steps_total"])
        out["steady_occupancy"] = out["queued_slot_steps_active"] / max(1, out["queued_slot_steps_total"])
        return out

    def last_timing(self):
        """(encoder_ms, cross_kv_ms, decode_ms, n_steps) of the last generate call (synchronises)."""
        arr = (C.c_float * 4)()
        _lib.check(self.lib.wseg_last_timing(self.handle, C.byref(arr)))
        return tuple(arr)
