// Epilogues shared by the GEMM kernels of wseg_gemm.hip (16-bit / split / mixed MFMA kernels, split-K reductions) and wseg_gemm_f32.hip
// (exact-parity fp32 kernels): 4 or 8 consecutive output columns of one row -> bias / GELU / residual / layout scatter.
#pragma once
#include <type_traits>
#include "wseg_kernels.h"

namespace wseg {

// ------------------------------------------------------------------------------------------------
// Epilogue: 4 consecutive columns n0..n0+3 of row m.
// ------------------------------------------------------------------------------------------------
template <typename T> struct Vec4 {      // 16-bit element types
  static __device__ __forceinline__ void ld(const T* p, float v[4]) {
    const uint2 t = *(const uint2*)p;
    v[0] = H16<T>::lo(t.x); v[1] = H16<T>::hi(t.x);
    v[2] = H16<T>::lo(t.y); v[3] = H16<T>::hi(t.y);
  }
  static __device__ __forceinline__ void st(T* p, const float v[4]) {
    uint2 t;
    t.x = H16<T>::pack(v[0], v[1]);
    t.y = H16<T>::pack(v[2], v[3]);
    *(uint2*)p = t;
  }
};
template <> struct Vec4<float> {
  static __device__ __forceinline__ void ld(const float* p, float v[4]) {
    const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  static __device__ __forceinline__ void st(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};

// Cross K / V as block floating point (EpiParams::kv24 == 2): the NL lanes that hold the 64 columns of one (position, head) row — 8
// consecutive lanes with 8 columns each (LDS-staged epilogues) or 16 with 4 each (split-K reduction) — agree on the row's power-of-two
// scale by DPP (the smallest 2^s with max|v| <= 2^15 * 2^s ... so that |v / 2^s| <= 32767 after the clamp), every lane stores its columns
// as int16 (round to nearest even) and the first lane the scale.  blk: the (slot, head) block [t_len][64] int16 + [t_len] float.
template <int NC>
__device__ __forceinline__ void st_bfp16_row(unsigned char* blk, int t_len, int t, int e, const float (&v)[NC], bool first_lane) {
  static_assert(NC == 4 || NC == 8, "4 or 8 columns per lane");
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) am = fmaxf(am, fabsf(v[i]));
  am = fmaxf(am, lane_xor<1>(am));
  am = fmaxf(am, lane_xor<2>(am));
  am = fmaxf(am, lane_xor<4>(am));
  if constexpr (NC == 4) am = fmaxf(am, lane_xor<8>(am));
  const unsigned bits = __float_as_uint(am);
  int ex = (int)(bits >> 23) - 127 + ((bits & 0x7fffffu) ? 1 : 0) - 15;      // ceil(log2 max) - 15
  ex = max(-110, min(ex, 110));
  const float inv = __uint_as_float((unsigned)(127 - ex) << 23), scl = __uint_as_float((unsigned)(127 + ex) << 23);
  int q[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) q[i] = max(-32767, min(32767, (int)__builtin_rintf(v[i] * inv)));
  unsigned w[NC / 2];
#pragma unroll
  for (int i = 0; i < NC / 2; ++i) w[i] = ((unsigned)q[2 * i] & 0xffffu) | ((unsigned)q[2 * i + 1] << 16);
  unsigned char* dst = blk + (size_t)t * 128 + e * 2;
  if constexpr (NC == 8) *(uint4*)dst = make_uint4(w[0], w[1], w[2], w[3]);
  else *(uint2*)dst = make_uint2(w[0], w[1]);
  if (first_lane) *(float*)(blk + (size_t)t_len * 128 + (size_t)t * 4) = scl;
}

// Cross K / V as block floating point with 24-BIT integers (EpiParams::kv24 == 3; r06): the same row agreement, elements as two's-complement
// 24-bit integers q with max|v| <= 2^23 * 2^s, stored in the two-plane layout of the 24-bit float format — [t_len][64] int16 (q >> 8), then
// [t_len][64] bytes (q & 0xff) — followed by [t_len] fp32 row scales 2^(s - 8): the reader rebuilds the 32-bit word [q >> 8 | q & 0xff | 0] =
// 256 q with the v_perm it already has, converts it with v_cvt_f32_i32 and multiplies the finished score / the probability by the stored
// scale.  Same bytes per row as the 24-bit floats (+ 4), error <= 2^-24 of the ROW maximum instead of 2^-17 of every element: the 24-bit
// floats were the largest term of f16x3's logit error (2e-5 of 2.5e-5) and cost it one of 4 200 sweep recordings.
template <int NC>
__device__ __forceinline__ void st_bfp24_row(unsigned char* blk, int t_len, int t, int e, const float (&v)[NC], bool first_lane) {
  static_assert(NC == 4 || NC == 8, "4 or 8 columns per lane");
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) am = fmaxf(am, fabsf(v[i]));
  am = fmaxf(am, lane_xor<1>(am));
  am = fmaxf(am, lane_xor<2>(am));
  am = fmaxf(am, lane_xor<4>(am));
  if constexpr (NC == 4) am = fmaxf(am, lane_xor<8>(am));
  const unsigned bits = __float_as_uint(am);
  int ex = (int)(bits >> 23) - 127 + ((bits & 0x7fffffu) ? 1 : 0) - 23;      // ceil(log2 max) - 23
  ex = max(-100, min(ex, 100));
  const float inv = __uint_as_float((unsigned)(127 - ex) << 23), scl = __uint_as_float((unsigned)(127 + ex - 8) << 23);
  int q[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) q[i] = max(-8388607, min(8388607, (int)__builtin_rintf(v[i] * inv)));
  unsigned hw[NC / 2];
#pragma unroll
  for (int i = 0; i < NC / 2; ++i) hw[i] = (((unsigned)q[2 * i] >> 8) & 0xffffu) | ((((unsigned)q[2 * i + 1] >> 8) & 0xffffu) << 16);
  unsigned lw[NC / 4];
#pragma unroll
  for (int i = 0; i < NC / 4; ++i)
    lw[i] = ((unsigned)q[4 * i] & 0xffu) | (((unsigned)q[4 * i + 1] & 0xffu) << 8) | (((unsigned)q[4 * i + 2] & 0xffu) << 16) | (((unsigned)q[4 * i + 3] & 0xffu) << 24);
  unsigned char* dh = blk + (size_t)t * 128 + e * 2;
  unsigned char* dl = blk + (size_t)t_len * 128 + (size_t)t * 64 + e;
  if constexpr (NC == 8) { *(uint4*)dh = make_uint4(hw[0], hw[1], hw[2], hw[3]); *(uint2*)dl = make_uint2(lw[0], lw[1]); }
  else { *(uint2*)dh = make_uint2(hw[0], hw[1]); *(unsigned*)dl = lw[0]; }
  if (first_lane) *(float*)(blk + (size_t)t_len * 192 + (size_t)t * 4) = scl;
}

// T: element-type tag of the mode (float | bf16_t | f16_t | X3<HT>); PT: its plain parameter type (bias, positional table,
// q / k / v storage: float in the split-precision modes).  Outputs that are the NEXT GEMM's operand (EPI_STORE, EPI_GELU) go
// through op_st*, i.e. as hi | lo pairs in the split-precision modes.
template <int EPI, typename T>
__device__ __forceinline__ void epi_apply(const EpiParams& ep, int m, int n0, float v[4]) {
  typedef typename IO<T>::P PT;
  if (ep.bias) {
    float b[4];
    Vec4<PT>::ld((const PT*)ep.bias + n0, b);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += b[i];
  }
  if constexpr (EPI == EPI_STORE) {
    op_st4<T>(ep.out, (size_t)m, ep.ldc, n0, v);
  } else if constexpr (EPI == EPI_GELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gelu_for<T>(v[i]);
    op_st4<T>(ep.out, (size_t)m, ep.ldc, n0, v);
  } else if constexpr (EPI == EPI_RESID) {      // the residual stream is fp32 in every mode
    float r[4];
    Vec4<float>::ld((const float*)ep.resid + (size_t)m * ep.ldc + n0, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = r[i] + v[i];
    Vec4<float>::st((float*)ep.out + (size_t)m * ep.ldc + n0, v);
  } else if constexpr (EPI == EPI_GELU_POS) {   // conv2 -> residual stream (fp32)
    float p[4];
    Vec4<PT>::ld((const PT*)ep.pos + (size_t)(m % ep.pos_rows) * ep.ldc + n0, p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gelu_for<T>(v[i]) + p[i];
    Vec4<float>::st((float*)ep.out + (size_t)m * ep.ldc + n0, v);
  } else if constexpr (EPI == EPI_QKV_ENC) {
    const int d = ep.d_model, sec = n0 / d, nn = n0 - sec * d, h = nn >> 6, e = nn & 63;
    const int b = m / ep.t_len, t = m - b * ep.t_len;
    const size_t bh = (size_t)b * ep.n_heads + h;
    if (sec == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= ep.scale;
    }
    auto put = [&](auto* base, size_t plane, const float* x) {
      typedef std::remove_pointer_t<decltype(base)> QT;
      if (sec == 0) Vec4<QT>::st((QT*)ep.q + plane + (bh * ep.t_pad + t) * 64 + e, x);
      else if (sec == 1) Vec4<QT>::st((QT*)ep.k + plane + (bh * ep.t_pad + t) * 64 + e, x);
      else if (ep.vt_tiled) {
        QT* vt = (QT*)ep.v + plane + bh * 64 * ep.t_pad;
#pragma unroll
        for (int i = 0; i < 4; ++i) El<QT>::st(vt + vt_tiled_index(e + i, t), x[i]);
      } else {
        QT* vt = (QT*)ep.v + plane + (bh * 64 + e) * ep.t_pad + t;
#pragma unroll
        for (int i = 0; i < 4; ++i) El<QT>::st(vt + (size_t)i * ep.t_pad, x[i]);
      }
    };
    typedef typename IO<T>::A AT;
    if (IO<T>::split && ep.qkv_mode == 1) put((float*)nullptr, 0, v);
    else {
      if constexpr (IO<T>::split) {                // the encoder attention's Q / K / V^T are IEEE-half planes in BOTH split modes:
#pragma unroll                                     // saturate like every other split operand (inf - inf = NaN in the lo plane otherwise)
        for (int i = 0; i < 4; ++i) v[i] = H16<AT>::sat(v[i]);
      }
      put((AT*)nullptr, 0, v);
      if (IO<T>::split && ep.qkv_mode == 2) {      // lo plane: x - rn(x)
        float lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lo[i] = v[i] - El<AT>::rnd(v[i]);
        put((AT*)nullptr, ep.qkv_plane, lo);
      }
    }
  } else if constexpr (EPI == EPI_KV_CROSS) {
    const int d = ep.d_model, sec = n0 / d, nn = n0 - sec * d, h = nn >> 6, e = nn & 63;
    const int b = m / ep.t_len, t = m - b * ep.t_len;
    const int bs = ep.slot_map ? ep.slot_map[b] : b;
    if (IO<T>::split && ep.kv24 == 2) {      // (reached from the split-K reduction only: 16 consecutive threads hold one row)
      unsigned char* blk = (unsigned char*)(sec == 0 ? ep.k : ep.v) + ((size_t)bs * ep.n_heads + h) * ep.t_len * 132;
      st_bfp16_row<4>(blk, ep.t_len, t, e, *(const float(*)[4])v, e == 0);
    } else if (IO<T>::split && ep.kv24 == 3) {
      unsigned char* blk = (unsigned char*)(sec == 0 ? ep.k : ep.v) + ((size_t)bs * ep.n_heads + h) * ep.t_len * 196;
      st_bfp24_row<4>(blk, ep.t_len, t, e, *(const float(*)[4])v, e == 0);
    } else if (IO<T>::split && ep.kv24) {
      unsigned char* blk = (unsigned char*)(sec == 0 ? ep.k : ep.v) + ((size_t)bs * ep.n_heads + h) * ep.t_len * 192;
      unsigned w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = __float_as_uint(v[i]) + 0x80u;        // round to 24 bits (half up in magnitude)
      *(uint2*)(blk + (size_t)t * 128 + e * 2) = make_uint2((w[0] >> 16) | (w[1] & 0xffff0000u), (w[2] >> 16) | (w[3] & 0xffff0000u));
      *(unsigned*)(blk + (size_t)ep.t_len * 128 + (size_t)t * 64 + e) =
          ((w[0] >> 8) & 0xffu) | (w[1] & 0xff00u) | ((w[2] << 8) & 0xff0000u) | ((w[3] << 16) & 0xff000000u);
    } else {
      PT* dst = (PT*)(sec == 0 ? ep.k : ep.v) + (((size_t)bs * ep.n_heads + h) * ep.t_len + t) * 64 + e;
      Vec4<PT>::st(dst, v);
    }
  } else if constexpr (EPI == EPI_F32) {
    *(float4*)(ep.out_f32 + (size_t)m * ep.ldc + n0) = make_float4(v[0], v[1], v[2], v[3]);
  } else if constexpr (EPI == EPI_QKV_DEC) {
    const int d = ep.d_model, sec = n0 / d, nn = n0 - sec * d, h = nn >> 6, e = nn & 63;
    if (sec == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] *= ep.scale;
      Vec4<PT>::st((PT*)ep.q + (size_t)m * d + nn, v);
    } else {
      const int slot = m / ep.pos_div, beam = m - slot * ep.pos_div;
      if (ep.idle_ptr[slot]) return;
      const int pos = ep.pos_ptr[slot];
      const int unit = ep.kv_pt[(size_t)slot * ep.kv_npg + pos / KV_PAGE];
      PT* dst = (PT*)(sec == 1 ? ep.k : ep.v) +
                ((((size_t)unit * ep.pos_div + beam) * ep.n_heads + h) * KV_PAGE + (pos % KV_PAGE)) * 64 + e;
      Vec4<PT>::st(dst, v);
    }
  } else if constexpr (EPI == EPI_SCALE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] *= ep.scale;
    Vec4<PT>::st((PT*)ep.out + (size_t)m * ep.ldc + n0, v);
  }
}

// 8 consecutive columns n0..n0+7 of row m (MFMA paths, LDS-staged epilogue): 16-byte loads / stores.
template <typename PT> __device__ __forceinline__ void ld8_h(const PT* p, float v[8]) {
  if constexpr (sizeof(PT) == 4) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
    unpack8<PT>(*(const uint4*)p, v);
  }
}
template <typename PT> __device__ __forceinline__ void st8_h(PT* p, const float v[8]) {
  if constexpr (sizeof(PT) == 4) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    *(uint4*)p = pack8<PT>(v);
  }
}

template <int EPI, typename T>
__device__ __forceinline__ void epi_apply8(const EpiParams& ep, int m, int n0, float v[8]) {
  typedef typename IO<T>::P PT;
  if (ep.bias) {
    float b[8];
    ld8_h<PT>((const PT*)ep.bias + n0, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
  }
  if constexpr (EPI == EPI_STORE) {
    op_st8<T>(ep.out, (size_t)m, ep.ldc, n0, v);
  } else if constexpr (EPI == EPI_GELU) {
    gelu8_for<T>(v);
    op_st8<T>(ep.out, (size_t)m, ep.ldc, n0, v);
  } else if constexpr (EPI == EPI_RESID) {      // the residual stream is fp32 in every mode
    const float* rp = (const float*)ep.resid + (size_t)m * ep.ldc + n0;
    const float4 r0 = *(const float4*)rp, r1 = *(const float4*)(rp + 4);
    float* o = (float*)ep.out + (size_t)m * ep.ldc + n0;
    *(float4*)o = make_float4(r0.x + v[0], r0.y + v[1], r0.z + v[2], r0.w + v[3]);
    *(float4*)(o + 4) = make_float4(r1.x + v[4], r1.y + v[5], r1.z + v[6], r1.w + v[7]);
  } else if constexpr (EPI == EPI_GELU_POS) {   // conv2 -> residual stream (fp32)
    float p[8];
    ld8_h<PT>((const PT*)ep.pos + (size_t)(m % ep.pos_rows) * ep.ldc + n0, p);
    gelu8_for<T>(v);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += p[i];
    float* o = (float*)ep.out + (size_t)m * ep.ldc + n0;
    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else if constexpr (EPI == EPI_QKV_ENC) {
    const int d = ep.d_model, sec = n0 / d, nn = n0 - sec * d, h = nn >> 6, e = nn & 63;
    const int b = m / ep.t_len, t = m - b * ep.t_len;
    const size_t bh = (size_t)b * ep.n_heads + h;
    if (sec == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] *= ep.scale;
    }
    auto put = [&](auto* base, size_t plane, const float* x) {
      typedef std::remove_pointer_t<decltype(base)> QT;
      if (sec == 0) st8_h<QT>((QT*)ep.q + plane + (bh * ep.t_pad + t) * 64 + e, x);
      else if (sec == 1) st8_h<QT>((QT*)ep.k + plane + (bh * ep.t_pad + t) * 64 + e, x);
      else if (ep.vt_tiled) {
        QT* vt = (QT*)ep.v + plane + bh * 64 * ep.t_pad;
#pragma unroll
        for (int i = 0; i < 8; ++i) El<QT>::st(vt + vt_tiled_index(e + i, t), x[i]);
      } else {
        QT* vt = (QT*)ep.v + plane + (bh * 64 + e) * ep.t_pad + t;
#pragma unroll
        for (int i = 0; i < 8; ++i) El<QT>::st(vt + (size_t)i * ep.t_pad, x[i]);
      }
    };
    typedef typename IO<T>::A AT;
    if (IO<T>::split && ep.qkv_mode == 1) put((float*)nullptr, 0, v);
    else {
      if constexpr (IO<T>::split) {                // IEEE-half planes in both split modes: saturate (see epi_apply)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = H16<AT>::sat(v[i]);
      }
      put((AT*)nullptr, 0, v);
      if (IO<T>::split && ep.qkv_mode == 2) {      // lo plane: x - rn(x)
        float lo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) lo[i] = v[i] - El<AT>::rnd(v[i]);
        put((AT*)nullptr, ep.qkv_plane, lo);
      }
    }
  } else if constexpr (EPI == EPI_KV_CROSS) {
    const int d = ep.d_model, sec = n0 / d, nn = n0 - sec * d, h = nn >> 6, e = nn & 63;
    const int b = m / ep.t_len, t = m - b * ep.t_len;
    const int bs = ep.slot_map ? ep.slot_map[b] : b;
    if (IO<T>::split && ep.kv24 == 2) {      // the 8 lanes of the row (LDS-staged epilogues: lane & 7 = column group)
      unsigned char* blk = (unsigned char*)(sec == 0 ? ep.k : ep.v) + ((size_t)bs * ep.n_heads + h) * ep.t_len * 132;
      st_bfp16_row<8>(blk, ep.t_len, t, e, *(const float(*)[8])v, e == 0);
    } else if (IO<T>::split && ep.kv24 == 3) {
      unsigned char* blk = (unsigned char*)(sec == 0 ? ep.k : ep.v) + ((size_t)bs * ep.n_heads + h) * ep.t_len * 196;
      st_bfp24_row<8>(blk, ep.t_len, t, e, *(const float(*)[8])v, e == 0);
    } else if (IO<T>::split && ep.kv24) {
      EpiParams e2 = ep;                        // the bias has been added above
      e2.bias = nullptr;
      epi_apply<EPI, T>(e2, m, n0, v);
      epi_apply<EPI, T>(e2, m, n0 + 4, v + 4);
    } else {
      PT* dst = (PT*)(sec == 0 ? ep.k : ep.v) + (((size_t)bs * ep.n_heads + h) * ep.t_len + t) * 64 + e;
      st8_h<PT>(dst, v);
    }
  } else if constexpr (EPI == EPI_F32) {
    float* o = ep.out_f32 + (size_t)m * ep.ldc + n0;
    *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    epi_apply<EPI, T>(ep, m, n0, v);
    epi_apply<EPI, T>(ep, m, n0 + 4, v + 4);
  }
}

}  // namespace wseg
