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

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// float -> bf16, round to nearest even: gfx950's v_cvt_pk_bf16_f32 (the integer emulation is ~7 VALU per element,
// which showed up as microseconds per 256x256 GEMM tile epilogue).
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const hw_f32x2 v = {lo, hi};
  const hw_bf16x2 b = __builtin_convertvector(v, hw_bf16x2);
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// Element-type traits: T = float (exact-parity mode) or bf16_t (production mode).
template <typename T> struct El;
template <> struct El<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct El<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// erf-GELU for the bf16 path: Abramowitz & Stegun 7.1.26 rational approximation of erf (|error| <= 1.5e-7, far
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
template <typename T> __device__ __forceinline__ float gelu_for(float x);
template <> __device__ __forceinline__ float gelu_for<float>(float x) { return gelu_erf(x); }
template <> __device__ __forceinline__ float gelu_for<uint16_t>(float x) { return gelu_erf_fast(x); }

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
