// Shared device/host helpers for libwseg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/wseg.h"

// Experiment knobs.  The PRODUCT library reads no tuning environment variable: the macros below fold to their defaults.  An A/B
// build keeps them live:  python -m whisperseg_amd.build --variant knobs -DWSEG_KNOBS=1  (tools/ab_variant.sh).  What the product
// does read are the TEST knobs that select an alternative kernel computing the same bits (WSEG_F32_GEMM, WSEG_F32_ATTN,
// WSEG_CROSS_NO_PK, WSEG_LOGMEL_GENERIC, WSEG_NO_GRAPH: tests compare both sides on the shipped library).
#ifdef WSEG_KNOBS
#define WSEG_KNOB_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define WSEG_KNOB_SET(name) (getenv(name) != nullptr)
#define WSEG_KNOB_IS(name, val) (getenv(name) && !strcmp(getenv(name), val))
#else
// (a tool that sets one of these variables against the product library would silently measure the default and label it an ablation:
// the library says so once per variable on stderr — ADVICE r05)
namespace wseg { bool knob_compiled_out(const char* name); }      // always false
#define WSEG_KNOB_INT(name, dflt) (::wseg::knob_compiled_out(name) ? (dflt) : (dflt))
#define WSEG_KNOB_SET(name) (::wseg::knob_compiled_out(name))
#define WSEG_KNOB_IS(name, val) (::wseg::knob_compiled_out(name))
#endif

namespace wseg {

void set_error(const char* fmt, ...);

#define WSEG_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::wseg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return WSEG_ERR_HIP;                                                                \
    }                                                                                     \
  } while (0)

#define WSEG_LAUNCH_CHECK()                                                               \
  do {                                                                                    \
    hipError_t _e = hipGetLastError();                                                    \
    if (_e != hipSuccess) {                                                               \
      ::wseg::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return WSEG_ERR_HIP;                                                                \
    }                                                                                     \
  } while (0)

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Two 16-bit storage / MFMA-operand types share every kernel: bfloat16 (WSEG_BF16) and IEEE half (WSEG_F16 — what the
// reference's own GPU fast path computes in: CTranslate2 compute_type "float16", reference model.py:691).  Both are raw
// 16-bit words in memory; f16_t is a distinct C++ type only so that templates can tell them apart.  Everything that knows
// the bit layout goes through H16<HT>: unpack / pack (round to nearest even) and the two MFMA shapes.
struct f16_t { uint16_t bits; };
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hw_f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 hw_f16x8 __attribute__((ext_vector_type(8)));

template <typename HT> struct H16;
template <> struct H16<bf16_t> {
  static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
  static __device__ __forceinline__ float one(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
  static __device__ __forceinline__ float sub_lo(float s, uint32_t w) { return s - lo(w); }
  static __device__ __forceinline__ float sub_hi(float s, uint32_t w) { return s - hi(w); }
  // gfx950's v_cvt_pk_bf16_f32 (the integer emulation is ~7 VALU per element, which showed up as microseconds per
  // 256x256 GEMM tile epilogue)
  static __device__ __forceinline__ uint32_t pack(float l, float h) {
    const hw_f32x2 v = {l, h};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
  }
  static __device__ __forceinline__ bf16_t from(float f) { return (bf16_t)(pack(f, 0.f) & 0xffffu); }
  static __device__ __forceinline__ float sat(float f) { return f; }      // bfloat16 has the fp32 exponent range
  static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct H16<f16_t> {
  static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(hw_f16x2, w)[0]; }
  static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(hw_f16x2, w)[1]; }
  static __device__ __forceinline__ float one(f16_t v) { return (float)__builtin_bit_cast(_Float16, v.bits); }
  // s - (float)half as ONE v_fma_mix_f32 (the half operand converted inside the instruction; fma(h, -1, s) rounds once, like the
  // subtraction: bit-identical) instead of v_cvt_f32_f16 + v_sub_f32 — the lo part of every hi + lo split
  static __device__ __forceinline__ float sub_lo(float s, uint32_t w) { return __builtin_fmaf(lo(w), -1.0f, s); }
  static __device__ __forceinline__ float sub_hi(float s, uint32_t w) { return __builtin_fmaf(hi(w), -1.0f, s); }
  // round to nearest even, IEEE overflow to infinity — as torch.float16 does.  The one tensor HF protects from that in
  // half precision is the residual stream (the clamp in modeling_whisper.py's encoder layer), which is fp32 here.
  static __device__ __forceinline__ uint32_t pack(float l, float h) {
    const hw_f32x2 v = {l, h};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_f16x2));
  }
  static __device__ __forceinline__ f16_t from(float f) { f16_t r; r.bits = (uint16_t)(pack(f, 0.f) & 0xffffu); return r; }
  // split-precision operands saturate at the largest finite half instead of becoming inf - inf = NaN (v_med3_f32); HF's own
  // half path clamps its hidden states the same way (modeling_whisper.py encoder layer)
  static __device__ __forceinline__ float sat(float f) { return __builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f); }
  static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hw_f16x8, a), __builtin_bit_cast(hw_f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hw_f16x8, a), __builtin_bit_cast(hw_f16x8, b), c, 0, 0, 0);
  }
};
__device__ __forceinline__ float bf2f(bf16_t v) { return H16<bf16_t>::one(v); }
__device__ __forceinline__ bf16_t f2bf(float f) { return H16<bf16_t>::from(f); }

// 8 consecutive 16-bit elements <-> 8 floats through one 16-byte access
template <typename HT> __device__ __forceinline__ void unpack8(const uint4& t, float v[8]) {
  const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) { v[2 * j] = H16<HT>::lo(w[j]); v[2 * j + 1] = H16<HT>::hi(w[j]); }
}
template <typename HT> __device__ __forceinline__ uint4 pack8(const float v[8]) {
  return make_uint4(H16<HT>::pack(v[0], v[1]), H16<HT>::pack(v[2], v[3]), H16<HT>::pack(v[4], v[5]), H16<HT>::pack(v[6], v[7]));
}

// V^T of the encoder self-attention in MFMA operand order (r06; EpiParams::vt_tiled): per (window, head) the [64 hd][t_pad keys] matrix is
// stored key tile by key tile — [t / 64][hd][64 keys] — and inside a 128-byte row the 16-byte slot 2 G + g2 holds the 8 keys that lane
// half g2 of v_mfma_f32_32x32x16 contracts for the 16-key group G: keys 16 G + 4 g2 + {0..3} | 16 G + 8 + 4 g2 + {0..3}.  The attention
// kernel then fetches a tile by LDS-DMA like the K tile and a fragment is one ds_read_b128.  Element offset inside the (window, head) block:
__host__ __device__ __forceinline__ int vt_tiled_index(int hd, int t) {
  const int k = t & 63, gi = (k >> 2) & 3;
  return (t >> 6) * 4096 + hd * 64 + ((2 * (k >> 4) + (gi & 1)) << 3) + ((gi >> 1) << 2) + (k & 3);
}

// Element-type traits: T = float (exact-parity mode), bf16_t or f16_t (production modes).
template <typename T> struct El {      // 16-bit types
  static __device__ __forceinline__ float ld(const T* p) { return H16<T>::one(*p); }
  static __device__ __forceinline__ void st(T* p, float v) { *p = H16<T>::from(v); }
  static __device__ __forceinline__ float rnd(float v) { return H16<T>::one(H16<T>::from(v)); }
};
template <> struct El<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};

// ------------------------------------------------------------------------------------------------------------------
// Split-precision ("x3") modes WSEG_BF16X3 / WSEG_F16X3.  Everything outside the GEMMs is the fp32 code of the exact-parity
// mode (fp32 Q / K / V, caches, softmax, LayerNorm, biases); a GEMM OPERAND value x — activations and weights — is carried as
// two 16-bit halves  hi = rn16(x), lo = rn16(x - hi)  (hi + lo == x to ~16 mantissa bits in bf16, ~22 in half) and a product
// sum is taken as  hi*hi + hi*lo + lo*hi  on the 16-bit matrix cores with fp32 accumulation (lo*lo, ~2^-16 / 2^-22 relative,
// is dropped).  Layout of an operand row of K logical elements: 2K 16-bit words, every 32 logical columns stored as
// [32 hi | 32 lo] (128 bytes).  A 64-word K tile of the 16-bit GEMM kernels is then exactly {hi, lo} of 32 logical columns,
// its two 32-wide MFMA k-steps are the hi and the lo fragments, and the x3 product is three MFMAs on fragments the kernel
// already holds: (W hi, A hi), (W hi, A lo), (W lo, A hi) — 3x the MFMAs and 2x the operand bytes of the plain 16-bit mode.
// X3<HT> is the element-type TAG of these modes (never instantiated); IO<T> maps a tag to its plain parameter type.
// ------------------------------------------------------------------------------------------------------------------
template <typename HT> struct X3 {};
struct f16_t;
// P: plain parameter / cache type; H: 16-bit MFMA operand type; A: storage type of the ENCODER self-attention's Q / K / V^T.
// Split-precision modes run the encoder attention on the IEEE-half matrix cores with fp32 softmax statistics — by default with
// Q, K, V^T and P as half hi + lo pairs and three MFMAs per product like the GEMMs (first-step logits within 5e-5 of the exact
// mode at 32 layers; plain half operands: 4e-4, which is what leaves the parity sweep untouched too — tools/precision_study.py
// "gemm=bf16x3,eattn=f16" 200 / 200 — but would make the attention the mode's largest error by 10x), whereas the decoder's
// cross-attention K / V must keep >= 16 mantissa bits ("ckv=f16": 188 / 200) and stay fp32.  WSEG_X3_ENC_ATTN=f16 | f32
// selects plain half / the fp32-MFMA kernel (EpiParams::qkv_mode).
template <typename T> struct IO { typedef T P; typedef T H; typedef T A; static constexpr bool split = false; };
template <typename HT> struct IO<X3<HT>> { typedef float P; typedef HT H; typedef f16_t A; static constexpr bool split = true; };

// ------------------------------------------------------------------------------------------------------------------
// Mixed split-precision mode WSEG_F16M6 (tag M6): everything OUTSIDE the GEMMs is the f16x3 code (fp32 arithmetic, producers write
// hi | lo IEEE-half operand rows); a GEMM multiplies  hi*hi  on the IEEE-half matrix cores and the two cross terms
// hi*lo + lo*hi  on the block-scaled MX matrix cores (v_mfma_scale_f32_16x16x128_f8f6f4, fp6 e2m3 operands: ~10 PFLOP/s dense, 4x
// the 16-bit rate) — the cross terms are 2^-11 of the product, so 3 mantissa bits + a per-32-element power-of-two scale keep the
// total operand error at ~2^-15.5 (bf16x3 class; tools/precision_study.py "gemm=f16m6": 200 / 200 sweep recordings identical).
// Per 64 logical columns: 2 half MFMAs (16 cycles each) + 1 MX MFMA (~20 cycles) instead of the six half MFMAs of f16x3.
// GEMM operands are "M6 rows" built from the hi | lo rows by x3_to_m6 (wseg_enc.hip): per 64 logical columns 256 bytes =
//   [64 hi halves (128 B)] [MX block (128 B)]      MX block = four 32-byte chunks, one per lane group g of the MX MFMA:
//   [24 bytes = 32 e2m3 codes][e8m0 scale byte][7 bytes padding].  ACTIVATION order: g = 0, 1 hold lo6 of logical columns 0..31 /
//   32..63 of the group, g = 2, 3 hold hi6 of the same columns; WEIGHT order: hi6 chunks first, then lo6 chunks.  Lane (row, g)
//   of the MX MFMA reads its chunk with two 16-byte LDS reads — the codes land in registers 0..5 of the operand, the scale byte
//   in register 6, which is passed as the instruction's scale operand — from BOTH operands: that pairs activation lo with weight
//   hi (g = 0, 1) and activation hi with weight lo (g = 2, 3): one instruction = both cross terms of 64 logical columns.
// A 64-word K tile of the GEMM kernels is alternately a hi tile (plain half MFMAs) and an MX tile.
// ------------------------------------------------------------------------------------------------------------------
struct M6 {};
template <> struct IO<M6> { typedef float P; typedef f16_t H; typedef f16_t A; static constexpr bool split = true; };
template <typename T> struct IsMx { static constexpr bool v = false; };
template <> struct IsMx<M6> { static constexpr bool v = true; };
typedef int mx_i32x8 __attribute__((ext_vector_type(8)));
struct MxFrag { mx_i32x8 v; };      // a 32-byte chunk: v[0..5] = 32 e2m3 codes, low byte of v[6] = e8m0 scale (the MFMA reads 6 registers)
__device__ __forceinline__ f32x4 mfma_mx6(const MxFrag& a, const MxFrag& b, f32x4 c) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, c, 2, 2, 0, a.v[6], 0, b.v[6]);
}

__host__ __device__ __forceinline__ int x3_col(int c) { return ((c >> 5) << 6) | (c & 31); }

// 8 consecutive logical columns (c % 8 == 0) <-> the two 16-byte pieces of a split row
template <typename HT> __device__ __forceinline__ void split8(const float v[8], uint4& hi, uint4& lo) {
  float s[8], r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = H16<HT>::sat(v[e]);
  hi = pack8<HT>(s);
  const uint32_t hw[4] = {hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[2 * e] = H16<HT>::sub_lo(s[2 * e], hw[e]); r[2 * e + 1] = H16<HT>::sub_hi(s[2 * e + 1], hw[e]); }
  lo = pack8<HT>(r);
}
// Operand stores / loads.  base: start of the operand matrix, ld: LOGICAL row length, c: logical column.
template <typename T> struct Op {        // plain element types (float / 16-bit)
  static __device__ __forceinline__ void st1(void* base, size_t row, int ld, int c, float v) { El<T>::st((T*)base + row * ld + c, v); }
  static __device__ __forceinline__ float ld1(const void* base, size_t row, int ld, int c) { return El<T>::ld((const T*)base + row * ld + c); }
};
template <typename HT> struct Op<X3<HT>> {
  static __device__ __forceinline__ void st1(void* base, size_t row, int ld, int c, float v) {
    uint16_t* p = (uint16_t*)base + row * (size_t)(2 * ld) + x3_col(c);
    v = H16<HT>::sat(v);
    const HT h = H16<HT>::from(v);
    const HT l = H16<HT>::from(v - H16<HT>::one(h));
    p[0] = __builtin_bit_cast(uint16_t, h);
    p[32] = __builtin_bit_cast(uint16_t, l);
  }
  static __device__ __forceinline__ float ld1(const void* base, size_t row, int ld, int c) {
    const uint16_t* p = (const uint16_t*)base + row * (size_t)(2 * ld) + x3_col(c);
    return H16<HT>::one(__builtin_bit_cast(HT, p[0])) + H16<HT>::one(__builtin_bit_cast(HT, p[32]));
  }
};
// ---- M6 rows written directly by a producer (activation order; layout above) ------------------------------------------------
// 32 IEEE halves held by ONE lane (16 registers of packed pairs) -> 24 bytes of e2m3 codes + the e8m0 scale byte of the block:
// the smallest power of two that brings the block's maximum inside e2m3's +-7.5 (nothing saturates), codes rounded to nearest even
// by v_cvt_scalef32_pk32_fp6_f16 (tools/probes/fp6_cvt_probe.hip pins its semantics and its agreement with the MX MFMA).
typedef _Float16 mx_h32 __attribute__((ext_vector_type(32)));
typedef unsigned mx_u6 __attribute__((ext_vector_type(6)));
__device__ __forceinline__ int mx_scale_byte(float amax) {      // smallest E with amax <= 7.5 * 2^E, as the e8m0 byte E + 127
  const float t = amax * (1.0f / 7.5f);
  int e = (int)((__float_as_uint(t) + 0x7fffffu) >> 23) - 127;      // ceil(log2 t) for normal t
  e = max(-120, min(e, 120));
  return e + 127;
}
// one 32-byte chunk of an MX block: [24 bytes codes][scale byte][7 bytes 0], as two 16-byte values
__device__ __forceinline__ void mx_chunk(const uint32_t (&w)[16], float amax, uint4& c0, uint4& c1) {
  union { uint32_t u[16]; mx_h32 h; } x;
#pragma unroll
  for (int i = 0; i < 16; ++i) x.u[i] = w[i];
  const int sb = mx_scale_byte(amax);
  const mx_u6 c = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x.h, __uint_as_float((unsigned)sb << 23));
  c0 = make_uint4(c[0], c[1], c[2], c[3]);
  c1 = make_uint4(c[4], c[5], (unsigned)sb, 0u);
}
// Cooperative store of 8 consecutive logical columns per lane: the FOUR lanes of an aligned quad (lane & 3 = 0..3) must call this
// together with c = c0, c0 + 8, c0 + 16, c0 + 24 (c0 % 32 == 0) of the SAME row — which is how every 8-column producer of the
// library is laid out (LayerNorm, the LDS-staged GEMM epilogues, the fused split-K reduction, the decoder attention outputs).
// Each lane writes its 8 hi halves; the quad shares its 16 + 16 packed words by DPP quad broadcasts, and lane k of the quad converts
// and writes piece k of {lo chunk first / second half, hi chunk first / second half}.
__device__ __forceinline__ void op_st8_m6(void* base, size_t row, int ld, int c, const float v[8]) {
  uint4 hi, lo;
  split8<f16_t>(v, hi, lo);
  unsigned char* blk = (unsigned char*)base + row * (size_t)(4 * ld) + (size_t)(c >> 6) * 256;      // the 256-byte block of 64 logical columns
  *(uint4*)(blk + (c & 63) * 2) = hi;
  float h8[8], l8[8];
  unpack8<f16_t>(hi, h8);
  unpack8<f16_t>(lo, l8);
  float ah = 0.f, al = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { ah = fmaxf(ah, fabsf(h8[e])); al = fmaxf(al, fabsf(l8[e])); }
  auto quad_max = [](float a) {      // maximum over the quad: quad_perm [1,0,3,2], then [2,3,0,1]
    a = fmaxf(a, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true)));
    return fmaxf(a, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true)));
  };
  ah = quad_max(ah);
  al = quad_max(al);
  // lanes 0, 1 of the quad store the two halves of the lo chunk, lanes 2, 3 those of the hi6 chunk: every lane gathers the 16 packed
  // words of ITS kind (quad broadcast k delivers lane k's word to all four lanes; a lane keeps the lo or the hi copy) and runs ONE
  // conversion
  const uint32_t hw[4] = {hi.x, hi.y, hi.z, hi.w}, lw[4] = {lo.x, lo.y, lo.z, lo.w};
  const int piece = (c >> 3) & 3;      // == lane & 3
  const bool want_hi = piece >= 2;
  uint32_t gw[16];
#define WSEG_QUAD_BCAST(K)                                                                                               \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                        \
    const uint32_t bh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hw[i], (K) * 0x55, 0xF, 0xF, true); /* quad_perm [K,K,K,K] */ \
    const uint32_t bl = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lw[i], (K) * 0x55, 0xF, 0xF, true);                \
    gw[4 * (K) + i] = want_hi ? bh : bl;                                                                                 \
  }
  WSEG_QUAD_BCAST(0) WSEG_QUAD_BCAST(1) WSEG_QUAD_BCAST(2) WSEG_QUAD_BCAST(3)
#undef WSEG_QUAD_BCAST
  uint4 c0, c1;
  mx_chunk(gw, want_hi ? ah : al, c0, c1);      // ONE conversion per lane: the chunk this lane stores half of
  const int chunk = (c >> 5) & 1;
  // (component-wise: the whole-vector select went through scratch in the GELU epilogue instantiations)
  const bool second = (piece & 1) != 0;
  const uint4 cs = make_uint4(second ? c1.x : c0.x, second ? c1.y : c0.y, second ? c1.z : c0.z, second ? c1.w : c0.w);
  *(uint4*)(blk + 128 + 32 * ((piece >> 1) * 2 + chunk) + 16 * (piece & 1)) = cs;
}

// The same row writer for producers that hold a strip of LDS (the LDS-staged GEMM epilogues): the quad exchanges its packed words
// through `quad` — 128 bytes of LDS owned by this aligned quad of lanes, [hi words of lanes 0..3 | lo words of lanes 0..3] — instead of
// 32 DPP broadcasts + 16 selects per lane, and the block maxima are taken on the fp32 values (rounding to half is monotone: the
// maximum of the rounded values IS the rounded maximum — bit-identical scales).  85 instead of 135 VALU instructions per 8 columns: the
// GELU epilogue of the encoder's fc1 GEMM is VALU-bound (all 8 waves of the workgroup write their tile at once, matrix pipe idle).
__device__ __forceinline__ void op_st8_m6_lds(void* base, size_t row, int ld, int c, const float v[8], unsigned char* quad, bool store) {
  float s[8], r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = H16<f16_t>::sat(v[e]);
  const uint4 hi = pack8<f16_t>(s);
  const uint32_t hw4[4] = {hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[2 * e] = H16<f16_t>::sub_lo(s[2 * e], hw4[e]); r[2 * e + 1] = H16<f16_t>::sub_hi(s[2 * e + 1], hw4[e]); }
  const uint4 lo = pack8<f16_t>(r);
  float ms = fabsf(s[0]), mr = fabsf(r[0]);
#pragma unroll
  for (int e = 1; e < 8; ++e) { ms = fmaxf(ms, fabsf(s[e])); mr = fmaxf(mr, fabsf(r[e])); }
  float ah = H16<f16_t>::lo(H16<f16_t>::pack(ms, 0.f)), al = H16<f16_t>::lo(H16<f16_t>::pack(mr, 0.f));
  auto quad_max = [](float a) {
    a = fmaxf(a, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true)));
    return fmaxf(a, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true)));
  };
  ah = quad_max(ah);
  al = quad_max(al);
  const int piece = (c >> 3) & 3;      // == lane & 3
  const bool want_hi = piece >= 2;
  *(uint4*)(quad + 16 * piece) = hi;
  *(uint4*)(quad + 64 + 16 * piece) = lo;
  // (LDS operations of one wave execute in order: the reads below see the quad's writes)
  const uint4* g = (const uint4*)(quad + (want_hi ? 0 : 64));
  const uint4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
  const uint32_t gw[16] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w};
  uint4 c0, c1;
  mx_chunk(gw, want_hi ? ah : al, c0, c1);
  if (!store) return;
  unsigned char* blk = (unsigned char*)base + row * (size_t)(4 * ld) + (size_t)(c >> 6) * 256;
  *(uint4*)(blk + (c & 63) * 2) = hi;
  const int chunk = (c >> 5) & 1;
  const bool second = (piece & 1) != 0;
  const uint4 cs = make_uint4(second ? c1.x : c0.x, second ? c1.y : c0.y, second ? c1.z : c0.z, second ? c1.w : c0.w);
  *(uint4*)(blk + 128 + 32 * ((piece >> 1) * 2 + chunk) + 16 * (piece & 1)) = cs;
}

template <typename T> __device__ __forceinline__ void op_st8(void* base, size_t row, int ld, int c, const float v[8]) {
  if constexpr (IsMx<T>::v) {
    op_st8_m6(base, row, ld, c, v);
  } else if constexpr (IO<T>::split) {
    typedef typename IO<T>::H HT;
    uint16_t* p = (uint16_t*)base + row * (size_t)(2 * ld) + x3_col(c);
    uint4 hi, lo;
    split8<HT>(v, hi, lo);
    *(uint4*)p = hi;
    *(uint4*)(p + 32) = lo;
  } else if constexpr (sizeof(T) == 4) {
    float* p = (float*)base + row * ld + c;
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    *(uint4*)((T*)base + row * ld + c) = pack8<T>(v);
  }
}
template <typename T> __device__ __forceinline__ void op_st4(void* base, size_t row, int ld, int c, const float v[4]) {
  if constexpr (IO<T>::split) {
    typedef typename IO<T>::H HT;
    uint16_t* p = (uint16_t*)base + row * (size_t)(2 * ld) + x3_col(c);
    const float s0 = H16<HT>::sat(v[0]), s1 = H16<HT>::sat(v[1]), s2 = H16<HT>::sat(v[2]), s3 = H16<HT>::sat(v[3]);
    uint2 hi = make_uint2(H16<HT>::pack(s0, s1), H16<HT>::pack(s2, s3));
    const float r0 = s0 - H16<HT>::lo(hi.x), r1 = s1 - H16<HT>::hi(hi.x), r2 = s2 - H16<HT>::lo(hi.y), r3 = s3 - H16<HT>::hi(hi.y);
    *(uint2*)p = hi;
    *(uint2*)(p + 32) = make_uint2(H16<HT>::pack(r0, r1), H16<HT>::pack(r2, r3));
  } else if constexpr (sizeof(T) == 4) {
    *(float4*)((float*)base + row * ld + c) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    *(uint2*)((T*)base + row * ld + c) = make_uint2(H16<T>::pack(v[0], v[1]), H16<T>::pack(v[2], v[3]));
  }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf-GELU for the 16-bit paths: Abramowitz & Stegun 7.1.26 rational approximation of erf (|error| <= 1.5e-7, far
// below the 2^-9 relative rounding of the bf16 result).  v_rcp_f32 / v_exp_f32 directly: `1.0f / x` and __frcp_rn
// compile to the IEEE division sequence (v_div_scale x2, v_div_fmas, v_div_fixup + Newton steps), which made the
// GELU of a 256x256 tile 3200 VALU instructions per wave.  12 VALU + 2 transcendental per element.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e2 = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);   // exp(-x^2 / 2)
  const float h = 0.5f * x;
  return fmaf(fabsf(h), fmaf(-(p * t), e2, 1.0f), h);    // 0.5 x (1 + sign(x) erf(|x| / sqrt 2))
}
// Two elements per instruction (v_pk_mul_f32 / v_pk_fma_f32 run two fp32 lanes-worth per issue): the same operations in the same
// order as gelu_erf_fast — bit-identical results, 15.5 instead of 20 issue slots per element.
typedef float gelu_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f2 gelu_erf_fast2(gelu_f2 x) {
  const gelu_f2 ax = {fabsf(x[0]), fabsf(x[1])};
  const gelu_f2 z = ax * (gelu_f2){0.70710678118654752440f, 0.70710678118654752440f};
  const gelu_f2 den = __builtin_elementwise_fma((gelu_f2){0.3275911f, 0.3275911f}, z, (gelu_f2){1.0f, 1.0f});
  const gelu_f2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  gelu_f2 p = __builtin_elementwise_fma((gelu_f2){1.061405429f, 1.061405429f}, t, (gelu_f2){-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, (gelu_f2){1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, (gelu_f2){-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, (gelu_f2){0.254829592f, 0.254829592f});
  const gelu_f2 a = (x * x) * (gelu_f2){-0.72134752044448170368f, -0.72134752044448170368f};
  const gelu_f2 e2 = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const gelu_f2 h = x * (gelu_f2){0.5f, 0.5f};
  const gelu_f2 ah = ax * (gelu_f2){0.5f, 0.5f};
  const gelu_f2 w = __builtin_elementwise_fma(-(p * t), e2, (gelu_f2){1.0f, 1.0f});
  return __builtin_elementwise_fma(ah, w, h);
}
template <typename T> __device__ __forceinline__ float gelu_for(float x) { return gelu_erf_fast(x); }      // 16-bit modes
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }
// eight elements of an 8-column epilogue: pairs through the packed form in the 16-bit / split modes (bit-identical to gelu_for, 4.5 issue
// slots per element fewer), the exact erff in f32 mode
#ifndef WSEG_GELU_PACKED
#define WSEG_GELU_PACKED 1      // A/B: build --variant sgelu -DWSEG_GELU_PACKED=0
#endif
template <typename T> __device__ __forceinline__ void gelu8_for(float* v) {
  if constexpr (WSEG_GELU_PACKED) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) { const gelu_f2 y = gelu_erf_fast2((gelu_f2){v[e], v[e + 1]}); v[e] = y[0]; v[e + 1] = y[1]; }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = gelu_erf_fast(v[e]);
  }
}
template <> __device__ __forceinline__ void gelu8_for<float>(float* v) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = gelu_erf(v[e]);
}

// Value of lane (l ^ M) of a wave64 WITHOUT the LDS crossbar: __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt, ~60-100 cycles
// of exposed latency each — the 96 of them that close a cross-attention workgroup were 5.7 of its 14 us at 8 slots (in-kernel
// s_memrealtime stamps).  M = 1, 2: DPP quad_perm; 4: row_shl:4 / row_shr:4 under bank masks; 8: row_ror:8; 16, 32: gfx950's
// v_permlane16_swap / v_permlane32_swap.  Every one is the exact exchange, so a + lane_xor<M>(a) is bit-equal to the shuffle form.
template <int M> __device__ __forceinline__ unsigned lane_xor_u(unsigned v) {
  static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "power of two below the wave size");
  const int x = (int)v;
  if constexpr (M == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);        // quad_perm [1,0,3,2]
  else if constexpr (M == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  else if constexpr (M == 4) {
    const int t = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xF, 0x5, false);                             // row_shl:4 -> banks 0, 2 (lanes 0-3, 8-11 take l + 4)
    return (unsigned)__builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);                          // row_shr:4 -> banks 1, 3 (lanes 4-7, 12-15 take l - 4)
  } else if constexpr (M == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, x, 0x128, 0xF, 0xF, true); // row_ror:8
  else if constexpr (M == 16) {
    // swap(first.row1, second.row0) per half: first -> {row0, row0}, second -> {row1, row1}; lane l of row r wants row r ^ 1
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return (__lane_id() & 16) ? r[0] : r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (__lane_id() & 32) ? r[0] : r[1];
  }
}
template <int M> __device__ __forceinline__ float lane_xor(float v) { return __uint_as_float(lane_xor_u<M>(__float_as_uint(v))); }
template <int M> __device__ __forceinline__ int lane_xor(int v) { return (int)lane_xor_u<M>((unsigned)v); }

__device__ __forceinline__ float wave_sum(float v) {
  v += lane_xor<32>(v); v += lane_xor<16>(v); v += lane_xor<8>(v); v += lane_xor<4>(v); v += lane_xor<2>(v); v += lane_xor<1>(v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, lane_xor<32>(v)); v = fmaxf(v, lane_xor<16>(v)); v = fmaxf(v, lane_xor<8>(v));
  v = fmaxf(v, lane_xor<4>(v)); v = fmaxf(v, lane_xor<2>(v)); v = fmaxf(v, lane_xor<1>(v));
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace wseg
