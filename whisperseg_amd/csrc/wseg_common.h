// Shared device/host helpers for libwseg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/wseg.h"

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
  // gfx950's v_cvt_pk_bf16_f32 (the integer emulation is ~7 VALU per element, which showed up as microseconds per
  // 256x256 GEMM tile epilogue)
  static __device__ __forceinline__ uint32_t pack(float l, float h) {
    const hw_f32x2 v = {l, h};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2));
  }
  static __device__ __forceinline__ bf16_t from(float f) { return (bf16_t)(pack(f, 0.f) & 0xffffu); }
  static __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct H16<f16_t> {
  static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(hw_f16x2, w)[0]; }
  static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(hw_f16x2, w)[1]; }
  static __device__ __forceinline__ float one(f16_t v) { return (float)__builtin_bit_cast(_Float16, v.bits); }
  // round to nearest even, IEEE overflow to infinity — as torch.float16 does.  The one tensor HF protects from that in
  // half precision is the residual stream (the clamp in modeling_whisper.py's encoder layer), which is fp32 here.
  static __device__ __forceinline__ uint32_t pack(float l, float h) {
    const hw_f32x2 v = {l, h};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_f16x2));
  }
  static __device__ __forceinline__ f16_t from(float f) { f16_t r; r.bits = (uint16_t)(pack(f, 0.f) & 0xffffu); return r; }
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
template <typename T> __device__ __forceinline__ float gelu_for(float x) { return gelu_erf_fast(x); }      // 16-bit modes
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace wseg
