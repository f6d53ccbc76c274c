// Encoder-side kernels for gfx950: conv-as-GEMM operand builders (im2col), LayerNorm, and the
// encoder self-attention (T = 500 keys, head_dim 64, no mask).
//
// Attention (bf16): flash-style, one workgroup = 128 query rows of one (window, head), 4 waves x 32 rows.
// Scores are computed TRANSPOSED (S^T = K Q^T with v_mfma_f32_32x32x16_bf16) so that a lane owns one
// query column: the softmax row reductions are 16 in-register values + one cross-half shuffle, and the
// exponentiated probabilities are already in the B-operand layout of the second MFMA
// (O^T = V^T P^T), so P never goes through LDS.  V^T is produced directly by the QKV GEMM epilogue
// (EPI_QKV_ENC), K and V^T tiles (64 keys) are staged in LDS with XOR swizzles that keep the
// ds_read_b128 / ds_read_b64 fragment reads conflict-free.
#include <stdlib.h>
#include <string.h>
#include "wseg_kernels.h"

namespace wseg {

// ------------------------------------------------------------------------------------------------
// im2col for conv1 (k=3, pad 1): A1[b*cols + t][tap*C + c] = x[b][c][t + tap - 1]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void im2col_conv1_kernel(const float* __restrict__ x, void* __restrict__ a1,
                                                           int C, int cols, int kp) {
  constexpr int TT = 32;  // time steps per block
  __shared__ float tile[96][TT + 2 + 1];
  const int b = blockIdx.y, t0 = blockIdx.x * TT;
  for (int i = threadIdx.x; i < C * (TT + 2); i += 256) {
    const int c = i / (TT + 2), j = i - c * (TT + 2);
    const int t = t0 + j - 1;
    tile[c][j] = (t >= 0 && t < cols) ? x[((size_t)b * C + c) * cols + t] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < TT * kp; i += 256) {
    const int tt = i / kp, k = i - tt * kp;
    if (t0 + tt >= cols) continue;
    float v = 0.f;
    if (k < 3 * C) {
      const int tap = k / C, c = k - tap * C;
      v = tile[c][tt + tap];
    }
    Op<T>::st1(a1, (size_t)b * cols + t0 + tt, kp, k, v);
  }
}

// im2col for conv2 (k=3, stride 2, pad 1): A2[b*(cols/2) + t][tap*d + c] = h1[b*cols + 2t + tap - 1][c]
template <typename T>
__global__ __launch_bounds__(256) void im2col_conv2_kernel(const T* __restrict__ h1, T* __restrict__ a2,
                                                           int B, int cols, int d) {
  constexpr int V = 16 / sizeof(T);
  const int dv = d / V, half = cols / 2;
  const size_t total = (size_t)B * half * 3 * dv;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cv = (int)(i % dv);
    const size_t r = i / dv;
    const int tap = (int)(r % 3);
    const size_t bt = r / 3;
    const int t = (int)(bt % half), b = (int)(bt / half);
    const int src = 2 * t + tap - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (src >= 0 && src < cols) v = *(const uint4*)(h1 + ((size_t)b * cols + src) * d + (size_t)cv * V);
    *(uint4*)(a2 + (bt * 3 + tap) * d + (size_t)cv * V) = v;
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm of the fp32 residual stream: one wave per row, 8 elements per lane and iteration (two 16-byte loads),
// two-pass statistics in registers, output in the model dtype TO (float in exact-parity mode).
// ------------------------------------------------------------------------------------------------
template <typename T> struct VecIO {      // 16-bit element types: 8 per 16-byte access
  static __device__ __forceinline__ void ld(const T* p, float v[8]) { unpack8<T>(*(const uint4*)p, v); }
  static __device__ __forceinline__ void st(T* p, const float v[8]) { *(uint4*)p = pack8<T>(v); }
};
template <> struct VecIO<float> {
  static __device__ __forceinline__ void ld(const float* p, float v[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void st(float* p, const float v[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};

template <typename TO, int NIT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const typename IO<TO>::P* __restrict__ g,
                                                        const typename IO<TO>::P* __restrict__ bta, void* __restrict__ y, int M, int d) {
  typedef typename IO<TO>::P PT;
  constexpr int V = 8;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * d;
  float v[NIT][V];
  float sum = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 64 + lane) * V;
    if (c < d) {
      VecIO<float>::ld(xr + c, v[it]);
#pragma unroll
      for (int j = 0; j < V; ++j) sum += v[it][j];
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) v[it][j] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)d;
  float sq = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 64 + lane) * V;
    if (c < d) {
#pragma unroll
      for (int j = 0; j < V; ++j) { const float t = v[it][j] - mean; sq += t * t; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)d + 1e-5f);
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int c = (it * 64 + lane) * V;
    if (c < d) {
      float gg[V], bb[V], o[V];
      VecIO<PT>::ld(g + c, gg);
      VecIO<PT>::ld(bta + c, bb);
#pragma unroll
      for (int j = 0; j < V; ++j) o[j] = (v[it][j] - mean) * rstd * gg[j] + bb[j];
      op_st8<TO>(y, (size_t)row, d, c, o);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Encoder attention, bf16 MFMA.
// ------------------------------------------------------------------------------------------------
// __launch_bounds__(256, 4): at most 128 registers, i.e. 4 instead of 3 workgroups per CU (847 -> 767 us per layer at 256
// windows; fewer VALU instructions at 2 per CU measured SLOWER: the kernel lives on occupancy).  Under that cap the
// register-prefetched K chunks were spilled to scratch directly behind their loads (PMC WRITE_SIZE showed 3.9x the
// algorithmic output bytes); with K prefetched by LDS-DMA instead the kernel has no scratch: 767 -> 571 us per layer.
// TO: tag of the output, the o-proj GEMM's operand (HT, or X3<..>: hi | lo rows in the split-precision modes)
// SPLIT (split-precision modes): Q, K and V^T arrive as hi + lo planes (`plane` elements apart), the probabilities are split in
// registers, and both products take three MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulation): S^T = K Q^T and O^T = V^T P^T
// to ~22 operand bits instead of 11.  Twice the LDS tiles and ~40 more registers: 3 workgroups per CU instead of 4.
template <typename HT, typename TO, bool SPLIT = false>
__global__ __launch_bounds__(256, SPLIT ? 3 : 4) void enc_attention_h16_kernel(const HT* __restrict__ Q, const HT* __restrict__ K,
                                                                const HT* __restrict__ Vt, void* __restrict__ out,
                                                                 int H, int T, int Tp, int d, size_t plane) {
  // K tile [key][64 hd] and V^T tile [hd][64 keys in MFMA operand order], 16-byte slots XOR ((row >> 1) & 7): a 32-row MFMA fragment read
  // (lane = row + 32 * half) is serviced in the lane groups {0-3,12-15,20-27}, ... and needs 16 distinct (row & 1, slot) pairs per group —
  // XOR (row & 7), right for 16-row fragments, was 2-way conflicted here (SQ_LDS_BANK_CONFLICT 44 % of LDS cycles).
  // Both tiles arrive by LDS-DMA (no registers in between; the swizzle is applied on the SOURCE side: LDS slot c of a row holds its global
  // slot (c & 7) ^ ((row >> 1) & 7)).  V^T (r06): the q | k | v GEMM epilogue writes it tile by tile in operand order — global
  // [b][h][key tile][hd][8 slots]: slot 2 G + g2 of a row holds the 8 keys lane half g2 contracts for the 16-key group G, keys
  // 16 G + 4 g2 + {0..3} | 16 G + 8 + 4 g2 + {0..3} (vt_tiled_index, wseg_common.h) — so a fragment is ONE ds_read_b128 and the tile needs no
  // register round trip (r02-r05: two 16-byte loads per thread into registers, four ds_write_b64 behind a barrier of their own).
  __shared__ __attribute__((aligned(16))) HT sK[2][64 * 64];   // double-buffered: K tile kt + 1 streams in under tile kt
  __shared__ __attribute__((aligned(16))) HT sV[64 * 64];      // single: V^T tile kt is requested when every wave is done with tile kt - 1
  __shared__ __attribute__((aligned(16))) HT sKl[SPLIT ? 2 : 1][SPLIT ? 64 * 64 : 8];      // lo planes (SPLIT)
  __shared__ __attribute__((aligned(16))) HT sVl[SPLIT ? 64 * 64 : 8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qi = lane & 31, g2 = lane >> 5;
  const HT* Qb = Q + (size_t)bh * Tp * 64;
  const HT* Kb = K + (size_t)bh * Tp * 64;
  const HT* Vb = Vt + (size_t)bh * 64 * Tp;

  bf16x8 qf[4], qfl[SPLIT ? 4 : 1];
#pragma unroll
  for (int hs = 0; hs < 4; ++hs) qf[hs] = *(const bf16x8*)(Qb + (size_t)(q0 + qi) * 64 + hs * 16 + g2 * 8);
  if constexpr (SPLIT) {
#pragma unroll
    for (int hs = 0; hs < 4; ++hs) qfl[hs] = *(const bf16x8*)(Qb + plane + (size_t)(q0 + qi) * 64 + hs * 16 + g2 * 8);
  }

  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1.0e30f, l_run = 0.f;

  const int n_tiles = (T + 63) / 64;
  // per-lane byte offset of the MFMA fragment reads of both tiles (see the key-block loop)
  const unsigned kfrag = (unsigned)(qi * 128 + ((g2 ^ ((qi >> 1) & 7)) << 4));
  // A tile is 8 KiB = 512 16-byte chunks, 2 per thread (rows are padded to Tp, so a tile is always readable).
  constexpr int NDMA = SPLIT ? 4 : 2;      // LDS-DMA instructions per thread and tile
  auto fetch_tile = [&](const HT* src, HT* dst, HT* dst_lo) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, row = c >> 3, sl = (c & 7) ^ ((row >> 1) & 7);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + row * 64 + sl * 8),
                                       (__attribute__((address_space(3))) void*)(dst + (i * 256 + wave * 64) * 8), 16, 0, 0);
      if constexpr (SPLIT)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + plane + row * 64 + sl * 8),
                                         (__attribute__((address_space(3))) void*)(dst_lo + (i * 256 + wave * 64) * 8), 16, 0, 0);
    }
  };
  auto fetch_k = [&](int kt) { fetch_tile(Kb + (size_t)(kt * 64) * 64, sK[kt & 1], sKl[kt & 1]); };
  auto fetch_v = [&](int kt) { fetch_tile(Vb + (size_t)kt * 64 * 64, sV, sVl); };
  // Hand-over protocol (two barriers per tile): barrier C at the top of tile kt — every wave's K chunks of tile kt have landed (they were
  // requested a whole tile earlier) and every wave is done with tile kt - 1, so sV and the other sK buffer may be refilled: V^T(kt) and
  // K(kt + 1) are requested; S^T = K Q^T and the softmax of the tile run while V^T(kt) lands; barrier D in front of P V.
  fetch_k(0);
  for (int kt = 0; kt < n_tiles; ++kt) {
    const HT* cK = sK[kt & 1];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's chunks of K(kt): the only request outstanding here
    __syncthreads();                                      // barrier C
    fetch_v(kt);
    const bool more = kt + 1 < n_tiles;
    if (more) fetch_k(kt + 1);
#ifndef WSEG_EA_PRIO
#define WSEG_EA_PRIO 1
#endif
    // Both 32-key halves of the tile together (r05): two independent score accumulators (the 12 MFMAs of one half are a dependent chain on ONE
    // accumulator: alternating halves lets the matrix pipe run back to back), one running maximum / rescale per 64 keys instead of per 32.
    // Measured and dropped (profiles/r05_encattn_ab.txt): the second half's probabilities computed between the first half's P V MFMAs (+20 %: VALU
    // inside a wave's MFMA stretch stalls both), the exponent arguments / row sums on v_pk_fma_f32 / v_pk_add_f32 (+1-2 %).
    {
      const int key_base = kt * 64;
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      // one lane offset per operand, made opaque once per key block so that its variants are v_xor'ed with a constant at the read instead of
      // living in loop-invariant registers (r05: three of them were spilled and reloaded behind an s_waitcnt vmcnt(0) in every key block)
      unsigned kfo = kfrag;
      asm volatile("" : "+v"(kfo));
      const char* cKs = (const char*)cK;
      [[maybe_unused]] const char* cKls = (const char*)sKl[kt & 1];
      // (three workgroups share a CU: a wave that has MFMAs to issue goes first — MI355X_MICROARCH.md, static priority — while its
      // neighbours' softmax VALU fills the slots between them)
      if (WSEG_EA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int hs = 0; hs < 4; ++hs) {
        const bf16x8 kf0 = *(const bf16x8*)(cKs + (kfo ^ (hs << 5)));
        const bf16x8 kf1 = *(const bf16x8*)(cKs + 4096 + (kfo ^ (hs << 5)));
        s0 = H16<HT>::mfma32(kf0, qf[hs], s0);
        s1 = H16<HT>::mfma32(kf1, qf[hs], s1);
        if constexpr (SPLIT) {
          const bf16x8 kfl0 = *(const bf16x8*)(cKls + (kfo ^ (hs << 5)));
          const bf16x8 kfl1 = *(const bf16x8*)(cKls + 4096 + (kfo ^ (hs << 5)));
          s0 = H16<HT>::mfma32(kf0, qfl[hs], s0);
          s1 = H16<HT>::mfma32(kf1, qfl[hs], s1);
          s0 = H16<HT>::mfma32(kfl0, qf[hs], s0);
          s1 = H16<HT>::mfma32(kfl1, qf[hs], s1);
        }
      }
      if (WSEG_EA_PRIO) __builtin_amdgcn_s_setprio(0);
      if (key_base + 64 > T) {      // the ragged tile of a (window, head): a real branch — if-converted, its key indices, compares and selects
        asm volatile("");           // ran in every key block
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key_base + (r & 3) + 8 * (r >> 2) + 4 * g2;
          if (key >= T) s0[r] = -1.0e30f;
          if (key + 32 >= T) s1[r] = -1.0e30f;
        }
      }
      float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);      // v_max3_f32
      mx = fmaxf(mx, lane_xor<32>(mx));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __expf(m_run - m_new);
      float ps = 0.f;
      // exp(s - m) = exp2(s log2e - m log2e): one v_fma + v_exp per probability (__expf: v_sub, v_mul, v_exp)
      const float m_l2 = -m_new * 1.4426950408889634f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], 1.4426950408889634f, m_l2));
        s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], 1.4426950408889634f, m_l2));
        ps += s0[r] + s1[r];
      }
      m_run = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {      // no query of this wave raised its maximum: 32 multiplies by 1.0 saved
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      }
      // this wave's chunks of V^T(kt) have landed (K(kt + 1), requested after them, stays in flight), then everyone's: barrier D
      // (a bare s_barrier, not __syncthreads: the fence of the latter would wait for K(kt + 1) as well)
      if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // O^T[hd][q] += V^T[hd][key] P^T[key][q]; the contraction slots of lane half g2 in MFMA (sub, mm) are the keys 32 sub + 16 mm + 4 g2 +
      // {0,1,2,3, 8,9,10,11} == registers 8 mm .. 8 mm + 7 of s<sub>: 16-key group G = 2 sub + mm of the operand-order V^T rows
      if (WSEG_EA_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const f32x16& sv = sub == 0 ? s0 : s1;
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          union { bf16x8 v; uint32_t u[4]; } pf, pfl;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) pf.u[jj] = H16<HT>::pack(sv[8 * mm + 2 * jj], sv[8 * mm + 2 * jj + 1]);
          if constexpr (SPLIT) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
              pfl.u[jj] = H16<HT>::pack(H16<HT>::sub_lo(sv[8 * mm + 2 * jj], pf.u[jj]), H16<HT>::sub_hi(sv[8 * mm + 2 * jj + 1], pf.u[jj]));
          }
#pragma unroll
          for (int ht = 0; ht < 2; ++ht) {
            const bf16x8 vf = *(const bf16x8*)((const char*)sV + ht * 4096 + (kfo ^ ((sub * 2 + mm) << 5)));      // the K fragment's address form
            if (ht == 0) o0 = H16<HT>::mfma32(vf, pf.v, o0);
            else o1 = H16<HT>::mfma32(vf, pf.v, o1);
            if constexpr (SPLIT) {
              const bf16x8 vfl = *(const bf16x8*)((const char*)sVl + ht * 4096 + (kfo ^ ((sub * 2 + mm) << 5)));
              if (ht == 0) { o0 = H16<HT>::mfma32(vf, pfl.v, o0); o0 = H16<HT>::mfma32(vfl, pf.v, o0); }
              else { o1 = H16<HT>::mfma32(vf, pfl.v, o1); o1 = H16<HT>::mfma32(vfl, pf.v, o1); }
            }
          }
        }
      }
      if (WSEG_EA_PRIO) __builtin_amdgcn_s_setprio(0);
      ps += lane_xor<32>(ps);
      l_run = l_run * alpha + ps;
    }
  }
  const int q = q0 + qi;
  if constexpr (IsMx<TO>::v) {
    // M6 rows (wseg_common.h): lane (qi, g2) holds, of query q and column half ht, the 16 columns 8 rg + 4 g2 + {0..3}; its partner
    // lane ^ 32 holds the other 16.  Lane g2 = 0 builds the 32-column block ht = 0 and lane g2 = 1 the block ht = 1: each splits
    // its 32 values into halves, hands the partner the 16 packed words (hi and lo) of the block the PARTNER builds
    // (v_permlane32_swap), and writes 64 bytes of hi halves + the lo6 and hi6 chunks of its block.
    const float inv = 1.0f / l_run;
    uint32_t hw[2][8], lw[2][8];                       // [ht][rg * 2 + pair]: packed halves of this lane's 16 columns of half ht
    float amax_h[2] = {0.f, 0.f}, amax_l[2] = {0.f, 0.f};
#pragma unroll
    for (int ht = 0; ht < 2; ++ht) {
      const f32x16& o = ht == 0 ? o0 : o1;
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const float a0 = H16<f16_t>::sat(o[2 * p] * inv), a1 = H16<f16_t>::sat(o[2 * p + 1] * inv);
        const uint32_t hh = H16<f16_t>::pack(a0, a1);
        const float r0 = a0 - H16<f16_t>::lo(hh), r1 = a1 - H16<f16_t>::hi(hh);
        const uint32_t ll = H16<f16_t>::pack(r0, r1);
        hw[ht][p] = hh; lw[ht][p] = ll;
        amax_h[ht] = fmaxf(amax_h[ht], fmaxf(fabsf(H16<f16_t>::lo(hh)), fabsf(H16<f16_t>::hi(hh))));
        amax_l[ht] = fmaxf(amax_l[ht], fmaxf(fabsf(H16<f16_t>::lo(ll)), fabsf(H16<f16_t>::hi(ll))));
      }
    }
    // my block = ht == g2; the partner needs my words of block 1 - g2
    uint32_t gh[16], gl[16];
    float ah = g2 ? amax_h[1] : amax_h[0], al = g2 ? amax_l[1] : amax_l[0];
    ah = fmaxf(ah, lane_xor<32>(g2 ? amax_h[0] : amax_h[1]));
    al = fmaxf(al, lane_xor<32>(g2 ? amax_l[0] : amax_l[1]));
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const uint32_t mine_h = g2 ? hw[1][p] : hw[0][p], give_h = g2 ? hw[0][p] : hw[1][p];
      const uint32_t mine_l = g2 ? lw[1][p] : lw[0][p], give_l = g2 ? lw[0][p] : lw[1][p];
      const uint32_t got_h = lane_xor_u<32>(give_h), got_l = lane_xor_u<32>(give_l);
      // columns of the block: 8 rg + 4 g2' + e; p = 2 rg + pair: the g2' = 0 lane's pair lands at word 4 rg + pair, the g2' = 1 lane's at 4 rg + 2 + pair
      const int rg = p >> 1, pr = p & 1;
      gh[4 * rg + pr] = g2 ? got_h : mine_h;
      gh[4 * rg + 2 + pr] = g2 ? mine_h : got_h;
      gl[4 * rg + pr] = g2 ? got_l : mine_l;
      gl[4 * rg + 2 + pr] = g2 ? mine_l : got_l;
    }
    if (q < T) {
      const size_t row = (size_t)b * T + q;
      const int c = h * 64 + g2 * 32;                  // first logical column of my block
      unsigned char* blk = (unsigned char*)out + row * (size_t)(4 * d) + (size_t)(c >> 6) * 256;
#pragma unroll
      for (int i = 0; i < 4; ++i) *(uint4*)(blk + (c & 63) * 2 + 16 * i) = make_uint4(gh[4 * i], gh[4 * i + 1], gh[4 * i + 2], gh[4 * i + 3]);
      uint4 c0, c1;
      const int chunk = (c >> 5) & 1;
      mx_chunk(gl, al, c0, c1);
      *(uint4*)(blk + 128 + 32 * chunk) = c0; *(uint4*)(blk + 128 + 32 * chunk + 16) = c1;
      mx_chunk(gh, ah, c0, c1);
      *(uint4*)(blk + 128 + 32 * (2 + chunk)) = c0; *(uint4*)(blk + 128 + 32 * (2 + chunk) + 16) = c1;
    }
  } else if (q < T) {
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int ht = 0; ht < 2; ++ht) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int hd = ht * 32 + 8 * rg + 4 * g2;
        const f32x16& o = ht == 0 ? o0 : o1;
        const float v4[4] = {o[4 * rg + 0] * inv, o[4 * rg + 1] * inv, o[4 * rg + 2] * inv, o[4 * rg + 3] * inv};
        op_st4<TO>(out, (size_t)b * T + q, d, h * 64 + hd, v4);
      }
    }
  }
}

// f32 exact-mode attention: one thread per query row (tests / tiny models only).
template <typename TO>      // float, or X3<HT>: the output is the o-proj GEMM's operand
__global__ __launch_bounds__(64) void enc_attention_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                               const float* __restrict__ Vt, void* __restrict__ out,
                                                               int H, int T, int Tp, int d) {
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int q = blockIdx.x * 64 + threadIdx.x;
  if (q >= T) return;
  const float* qp = Q + ((size_t)bh * Tp + q) * 64;
  float qv[64], o[64];
#pragma unroll
  for (int e = 0; e < 64; ++e) { qv[e] = qp[e]; o[e] = 0.f; }
  // pass 1: row max; pass 2: exp / sum / PV  (same order of operations as softmax(QK^T) V)
  float mx = -3.0e38f;
  for (int t = 0; t < T; ++t) {
    const float* kp = K + ((size_t)bh * Tp + t) * 64;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 64; ++e) s = fmaf(qv[e], kp[e], s);
    mx = fmaxf(mx, s);
  }
  float l = 0.f;
  for (int t = 0; t < T; ++t) {
    const float* kp = K + ((size_t)bh * Tp + t) * 64;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 64; ++e) s = fmaf(qv[e], kp[e], s);
    const float p = expf(s - mx);
    l += p;
    const float* vp = Vt + (size_t)bh * 64 * Tp + t;
#pragma unroll
    for (int e = 0; e < 64; ++e) o[e] = fmaf(p, vp[(size_t)e * Tp], o[e]);
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int e = 0; e < 64; e += 8) {
    float v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v8[u] = o[e + u] * inv;
    op_st8<TO>(out, (size_t)b * T + q, d, h * 64 + e, v8);
  }
}

// The same arithmetic on the fp32 matrix cores, bit for bit: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain, so
//   S^T[t][q] = sum_e K[t][e] Q[q][e]      (chain over e = 0..63, as the loop above)
//   l[q]      = sum_t 1 * P^T[t][q]        (fma(1, p, l) == l + p: the ascending-t sum above)
//   O^T[e][q] = sum_t V^T[e][t] P^T[t][q]  (chain over t ascending)
// with the row maximum taken in a first pass over the keys exactly as above.  Keys past T get p = 0 exactly, which leaves
// both chains unchanged (K / V^T pad rows are finite).  One wave per 32 queries, 4 waves per workgroup share the K / V^T
// tiles of 32 keys in LDS; P^T goes through the wave's own LDS tile to reach the MFMA operand layout.
template <typename TO>
__global__ __launch_bounds__(256) void enc_attention_f32_mfma_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                                     const float* __restrict__ Vt, void* __restrict__ out,
                                                                     int H, int T, int Tp, int d) {
  __shared__ float sK[32][65];          // [key][hd]
  __shared__ float sV[64][33];          // [hd][key]
  __shared__ float sP[4][32][33];       // per wave: [key][query]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fi = lane & 31, fk = lane >> 5;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const float* Qb = Q + (size_t)bh * Tp * 64;
  const float* Kb = K + (size_t)bh * Tp * 64;
  const float* Vb = Vt + (size_t)bh * 64 * Tp;
  const int qrow = min(q0 + fi, Tp - 1);
  float qf[32];                         // B operand of S^T = K Q^T: Q[q0 + fi][2 kk + fk]
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) qf[kk] = Qb[(size_t)qrow * 64 + 2 * kk + fk];
  const int n_tiles = (T + 31) / 32;
  auto load_k = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, row = c >> 4, col = (c & 15) * 4;
      const float4 v = *(const float4*)(Kb + (size_t)(kt * 32 + row) * 64 + col);
      sK[row][col] = v.x; sK[row][col + 1] = v.y; sK[row][col + 2] = v.z; sK[row][col + 3] = v.w;
    }
  };
  auto load_v = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, row = c >> 3, col = (c & 7) * 4;
      const float4 v = *(const float4*)(Vb + (size_t)row * Tp + kt * 32 + col);
      sV[row][col] = v.x; sV[row][col + 1] = v.y; sV[row][col + 2] = v.z; sV[row][col + 3] = v.w;
    }
  };
  auto scores = [&](int kt, f32x16& sc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[fi][2 * kk + fk], qf[kk], sc, 0, 0, 0);
  };
  // pass 1: row maximum over the real keys
  float mx = -3.0e38f;
  for (int kt = 0; kt < n_tiles; ++kt) {
    __syncthreads();
    load_k(kt);
    __syncthreads();
    f32x16 sc;
    scores(kt, sc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + 8 * (r >> 2) + 4 * fk + (r & 3);
      if (key < T) mx = fmaxf(mx, sc[r]);
    }
  }
  mx = fmaxf(mx, lane_xor<32>(mx));       // the two lane halves hold the two key halves of a query
  // pass 2: p = exp(s - max), l += p, O += p V — all in ascending key order
  f32x16 o0, o1, la;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; la[r] = 0.f; }
  for (int kt = 0; kt < n_tiles; ++kt) {
    __syncthreads();
    load_k(kt);
    load_v(kt);
    __syncthreads();
    f32x16 sc;
    scores(kt, sc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kl = 8 * (r >> 2) + 4 * fk + (r & 3);
      sP[wave][kl][fi] = (kt * 32 + kl < T) ? expf(sc[r] - mx) : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the wave's own LDS writes, then its own reads
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float pb = sP[wave][2 * kk + fk][fi];
      o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[fi][2 * kk + fk], pb, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[32 + fi][2 * kk + fk], pb, o1, 0, 0, 0);
      la = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, pb, la, 0, 0, 0);
    }
  }
  const int q = q0 + fi;
  if (q < T) {
    const float inv = 1.0f / la[0];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int hd = 8 * g + 4 * fk;
      const float a[4] = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
      const float c[4] = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
      op_st4<TO>(out, (size_t)b * T + q, d, h * 64 + hd, a);
      op_st4<TO>(out, (size_t)b * T + q, d, h * 64 + 32 + hd, c);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------
static bool is_x3(int dtype) { return dtype == WSEG_BF16X3 || dtype == WSEG_F16X3; }
// attribution knob (tools/parity_sweep.py, profiles/): fp32-MFMA encoder attention in the split-precision modes
int x3_enc_attention_mode() {
  static const int v = WSEG_KNOB_IS("WSEG_X3_ENC_ATTN", "f32") ? 1 : (WSEG_KNOB_IS("WSEG_X3_ENC_ATTN", "f16") ? 0 : 2);      // (variant builds)
  return v;
}

// Split-precision operand rows [M][2d words] <-> fp32 [M][d] (the C-ABI hands encoder states over as fp32 in these modes).
template <typename HT>
__global__ __launch_bounds__(256) void operand_to_f32_kernel(const uint16_t* __restrict__ op, float* __restrict__ out, size_t n8, int d) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const size_t row = i / (d >> 3);
    const int c = (int)(i - row * (d >> 3)) << 3;
    const uint16_t* p = op + row * (size_t)(2 * d) + x3_col(c);
    float h[8], l[8];
    unpack8<HT>(*(const uint4*)p, h);
    unpack8<HT>(*(const uint4*)(p + 32), l);
    float* o = out + row * d + c;
    *(float4*)o = make_float4(h[0] + l[0], h[1] + l[1], h[2] + l[2], h[3] + l[3]);
    *(float4*)(o + 4) = make_float4(h[4] + l[4], h[5] + l[5], h[6] + l[6], h[7] + l[7]);
  }
}
template <typename HT>
__global__ __launch_bounds__(256) void f32_to_operand_kernel(const float* __restrict__ in, void* __restrict__ op, size_t n8, int d) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const size_t row = i / (d >> 3);
    const int c = (int)(i - row * (d >> 3)) << 3;
    float v[8];
    VecIO<float>::ld(in + row * d + c, v);
    op_st8<X3<HT>>(op, row, d, c, v);
  }
}
// ------------------------------------------------------------------------------------------------
// WSEG_F16M6: hi | lo IEEE-half operand rows -> M6 rows (layout: wseg_common.h).  One lane per 32 logical columns (one 128-byte
// [32 hi | 32 lo] group of the source row): the block maxima of |hi| and |lo| give the two e8m0 scales (the smallest power of two
// that brings the block inside e2m3's +-7.5: nothing saturates), v_cvt_scalef32_pk32_fp6_f16 quantises the 32 halves of a lane to
// 24 bytes of e2m3 codes (round to nearest even; tools/probes/fp6_cvt_probe.hip pins its semantics and its agreement with the MX
// MFMA), the hi halves are copied.  The wave's 8 KB of source rows arrive by fully coalesced 16-byte loads through a padded LDS
// image (144-byte pitch: conflict-free 16-byte reads at a 128-byte lane stride).
// ------------------------------------------------------------------------------------------------
template <bool WORDER>
__global__ __launch_bounds__(256) void x3_to_m6_kernel(const uint16_t* __restrict__ src, unsigned char* __restrict__ dst, size_t n_groups, int K) {
  __shared__ __attribute__((aligned(16))) unsigned char stage[4][64 * 144];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gpr = K >> 5;                                   // 32-column groups per row
  for (size_t g0 = ((size_t)blockIdx.x * 4 + wave) * 64; g0 < n_groups; g0 += (size_t)gridDim.x * 256) {
    // groups g0 .. g0 + 63 are 8 KB of contiguous source bytes (a row is gpr groups of 128 bytes, rows are contiguous)
    const unsigned char* sp = (const unsigned char*)src + g0 * 128;
    const size_t n_here = n_groups - g0 < 64 ? n_groups - g0 : 64;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int piece = q * 64 + lane;                        // 16-byte piece of the 8 KB
      const int grp = piece >> 3;
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((size_t)grp < n_here) v = *(const uint4*)(sp + (size_t)piece * 16);
      *(uint4*)(stage[wave] + grp * 144 + (piece & 7) * 16) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    union { uint4 q[4]; mx_h32 h; } hi, lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) { hi.q[q] = *(const uint4*)(stage[wave] + lane * 144 + q * 16); lo.q[q] = *(const uint4*)(stage[wave] + lane * 144 + 64 + q * 16); }
    __builtin_amdgcn_wave_barrier();                          // every lane has read its group before the image is overwritten
    if ((size_t)lane < n_here) {
      float ah = 0.f, al = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) { ah = fmaxf(ah, fabsf((float)hi.h[i])); al = fmaxf(al, fabsf((float)lo.h[i])); }
      const int sbh = mx_scale_byte(ah), sbl = mx_scale_byte(al);
      const mx_u6 ch = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(hi.h, __uint_as_float((unsigned)sbh << 23));
      const mx_u6 cl = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(lo.h, __uint_as_float((unsigned)sbl << 23));
      const size_t g = g0 + lane, row = g / gpr;
      const int gi = (int)(g - row * gpr), chunk = gi & 1;
      unsigned char* d64 = dst + row * (size_t)(4 * K) + (size_t)(gi >> 1) * 256;      // the 256-byte block of 64 logical columns
#pragma unroll
      for (int q = 0; q < 4; ++q) *(uint4*)(d64 + chunk * 64 + q * 16) = hi.q[q];
      // chunk g of the MX block: activation order lo chunks 0, 1 then hi chunks 0, 1; weight order hi first
      unsigned char* mx = d64 + 128;
      unsigned char* plo = mx + 32 * ((WORDER ? 2 : 0) + chunk);
      unsigned char* phi = mx + 32 * ((WORDER ? 0 : 2) + chunk);
      *(uint4*)plo = make_uint4(cl[0], cl[1], cl[2], cl[3]);
      *(uint4*)(plo + 16) = make_uint4(cl[4], cl[5], (unsigned)sbl, 0u);
      *(uint4*)phi = make_uint4(ch[0], ch[1], ch[2], ch[3]);
      *(uint4*)(phi + 16) = make_uint4(ch[4], ch[5], (unsigned)sbh, 0u);
    }
  }
}

int launch_x3_to_m6(const void* x3_rows, void* m6_rows, size_t M, int K, bool weight_order, hipStream_t s) {
  if (K <= 0 || K % 64) { set_error("x3_to_m6: K %d must be a multiple of 64", K); return WSEG_ERR_INVALID; }
  const size_t n_groups = M * (size_t)(K >> 5);
  if (n_groups == 0) return WSEG_OK;
  const size_t want = (n_groups + 255) / 256;
  const int blocks = (int)(want < 16384 ? want : 16384);
  if (weight_order) hipLaunchKernelGGL(x3_to_m6_kernel<true>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)x3_rows, (unsigned char*)m6_rows, n_groups, K);
  else hipLaunchKernelGGL(x3_to_m6_kernel<false>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)x3_rows, (unsigned char*)m6_rows, n_groups, K);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

int launch_operand_to_f32(int dtype, const void* op, float* out, size_t M, int d, hipStream_t s) {
  if (!is_x3(dtype) || d % 32) { set_error("operand_to_f32: dtype %d / d %d unsupported", dtype, d); return WSEG_ERR_INVALID; }
  const size_t n8 = M * (size_t)(d >> 3);
  if (n8 == 0) return WSEG_OK;
  const int blocks = (int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
  if (dtype == WSEG_BF16X3) hipLaunchKernelGGL(operand_to_f32_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)op, out, n8, d);
  else hipLaunchKernelGGL(operand_to_f32_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, (const uint16_t*)op, out, n8, d);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_f32_to_operand(int dtype, const float* in, void* op, size_t M, int d, hipStream_t s) {
  if (!is_x3(dtype) || d % 32) { set_error("f32_to_operand: dtype %d / d %d unsupported", dtype, d); return WSEG_ERR_INVALID; }
  const size_t n8 = M * (size_t)(d >> 3);
  if (n8 == 0) return WSEG_OK;
  const int blocks = (int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
  if (dtype == WSEG_BF16X3) hipLaunchKernelGGL(f32_to_operand_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, in, op, n8, d);
  else hipLaunchKernelGGL(f32_to_operand_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, in, op, n8, d);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

int launch_im2col_conv1(int dtype, const float* feats, void* a1, int B, int n_mels, int cols, int kp, hipStream_t s) {
  if (n_mels > 96 || 3 * n_mels > kp) { set_error("im2col_conv1: n_mels %d unsupported", n_mels); return WSEG_ERR_INVALID; }
  dim3 grid(cdiv(cols, 32), B);
#define WSEG_IC1(T_) hipLaunchKernelGGL((im2col_conv1_kernel<T_>), grid, dim3(256), 0, s, feats, a1, n_mels, cols, kp)
  if (dtype == WSEG_BF16) WSEG_IC1(bf16_t);
  else if (dtype == WSEG_F16) WSEG_IC1(f16_t);
  else if (dtype == WSEG_BF16X3) WSEG_IC1(X3<bf16_t>);
  else if (dtype == WSEG_F16X3) WSEG_IC1(X3<f16_t>);
  else WSEG_IC1(float);
#undef WSEG_IC1
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

int launch_im2col_conv2(int dtype, const void* h1, void* a2, int B, int cols, int d, hipStream_t s) {
  // split-precision rows are 2d 16-bit words per tap, and x3_col() is 32-column-blockwise (d % 32 == 0): the split image of
  // A2 row [tap0 | tap1 | tap2] is the three split h1 rows back to back — the same 16-byte copy with d -> 2d
  if (is_x3(dtype)) d *= 2;
  const size_t vec = dtype == WSEG_F32 ? 4 : 8;
  const size_t total = (size_t)B * (cols / 2) * 3 * (d / vec);
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype != WSEG_F32) hipLaunchKernelGGL((im2col_conv2_kernel<bf16_t>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)h1, (bf16_t*)a2, B, cols, d);   // a pure 16-byte copy
  else hipLaunchKernelGGL((im2col_conv2_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)h1, (float*)a2, B, cols, d);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

int launch_layernorm(int dtype, const float* x, const void* g, const void* b, void* y, int M, int d, hipStream_t s) {
  if (M <= 0) return WSEG_OK;
  dim3 grid(cdiv(M, 4));
  if (d % 8 || d > 64 * 8 * 4) { set_error("layernorm: d %d unsupported", d); return WSEG_ERR_INVALID; }
#define WSEG_LN(TO_, NIT_) hipLaunchKernelGGL((layernorm_kernel<TO_, NIT_>), grid, dim3(256), 0, s, x, (const typename IO<TO_>::P*)g, (const typename IO<TO_>::P*)b, y, M, d)
#define WSEG_LN3(TO_) do { if (d <= 512) WSEG_LN(TO_, 1); else if (d <= 1024) WSEG_LN(TO_, 2); else WSEG_LN(TO_, 4); } while (0)
  if (dtype == WSEG_BF16) WSEG_LN3(bf16_t);
  else if (dtype == WSEG_F16) WSEG_LN3(f16_t);
  else if (dtype == WSEG_BF16X3) WSEG_LN3(X3<bf16_t>);
  else if (dtype == WSEG_F16X3) WSEG_LN3(X3<f16_t>);
  else if (dtype == WSEG_F16M6) WSEG_LN3(M6);      // M6 rows written directly (the four lanes of a quad share a 32-column block)
  else WSEG_LN3(float);
#undef WSEG_LN3
#undef WSEG_LN
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

template <typename TO>
static void launch_enc_attention_f32(const void* q, const void* k, const void* vt, void* out, int B, int H, int T, int Tp, int d, hipStream_t s) {
  static const bool naive = getenv("WSEG_F32_ATTN") && !strcmp(getenv("WSEG_F32_ATTN"), "naive");   // test knob: same bits either way
  if (naive || Tp % 128 != 0) {
    dim3 grid(cdiv(T, 64), B * H);
    hipLaunchKernelGGL(enc_attention_f32_kernel<TO>, grid, dim3(64), 0, s, (const float*)q, (const float*)k, (const float*)vt, out, H, T, Tp, d);
  } else {
    dim3 grid(cdiv(T, 128), B * H);
    hipLaunchKernelGGL(enc_attention_f32_mfma_kernel<TO>, grid, dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)vt, out, H, T, Tp, d);
  }
}

bool enc_attention_writes_mx(int dtype) { return dtype == WSEG_F16M6 && x3_enc_attention_mode() == 2; }
bool enc_attention_vt_tiled(int dtype) {
  if (dtype == WSEG_F32) return false;
  const bool split = dtype == WSEG_BF16X3 || dtype == WSEG_F16X3 || dtype == WSEG_F16M6;
  return !(split && x3_enc_attention_mode() == 1);
}

int launch_enc_attention(int dtype, const void* q, const void* k, const void* vt, void* out,
                         int B, int H, int T, int Tp, int d, hipStream_t s) {
  if (dtype == WSEG_F16M6 && x3_enc_attention_mode() == 2) {      // split-precision attention writing M6 rows directly
    if (Tp % 128) { set_error("enc_attention: Tp %d %% 128", Tp); return WSEG_ERR_INVALID; }
    dim3 grid(cdiv(T, 128), B * H);
    hipLaunchKernelGGL((enc_attention_h16_kernel<f16_t, M6, true>), grid, dim3(256), 0, s, (const f16_t*)q, (const f16_t*)k, (const f16_t*)vt, out, H, T, Tp, d,
                       (size_t)B * H * Tp * 64);
    WSEG_LAUNCH_CHECK();
    return WSEG_OK;
  }
  dtype = storage_dtype(dtype);
  const bool x3 = is_x3(dtype);
  const int mode = x3 ? x3_enc_attention_mode() : 0;
  if (dtype == WSEG_BF16 || dtype == WSEG_F16 || (x3 && mode != 1)) {
    if (Tp % 128) { set_error("enc_attention: Tp %d %% 128", Tp); return WSEG_ERR_INVALID; }
    dim3 grid(cdiv(T, 128), B * H);
    const size_t plane = (size_t)B * H * Tp * 64;
#define WSEG_EA(HT_, TO_, SP_) hipLaunchKernelGGL((enc_attention_h16_kernel<HT_, TO_, SP_>), grid, dim3(256), 0, s, (const HT_*)q, (const HT_*)k, (const HT_*)vt, out, H, T, Tp, d, plane)
    if (dtype == WSEG_BF16) WSEG_EA(bf16_t, bf16_t, false);
    else if (dtype == WSEG_F16) WSEG_EA(f16_t, f16_t, false);
    else if (dtype == WSEG_BF16X3) { if (mode == 2) WSEG_EA(f16_t, X3<bf16_t>, true); else WSEG_EA(f16_t, X3<bf16_t>, false); }   // IEEE-half
    else { if (mode == 2) WSEG_EA(f16_t, X3<f16_t>, true); else WSEG_EA(f16_t, X3<f16_t>, false); }    // Q / K / V^T in both split modes
#undef WSEG_EA
  } else if (dtype == WSEG_BF16X3) launch_enc_attention_f32<X3<bf16_t>>(q, k, vt, out, B, H, T, Tp, d, s);
  else if (dtype == WSEG_F16X3) launch_enc_attention_f32<X3<f16_t>>(q, k, vt, out, B, H, T, Tp, d, s);
  else launch_enc_attention_f32<float>(q, k, vt, out, B, H, T, Tp, d, s);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

}  // namespace wseg

extern "C" int wseg_convert_operand(const void* src_x3_rows, void* dst_m6_rows, int64_t n_rows, int32_t K, int32_t weight_order, void* stream) {
  using namespace wseg;
  if (!src_x3_rows || !dst_m6_rows || n_rows < 0 || src_x3_rows == dst_m6_rows) { set_error("wseg_convert_operand: bad argument"); return WSEG_ERR_INVALID; }
  return launch_x3_to_m6(src_x3_rows, dst_m6_rows, (size_t)n_rows, K, weight_order != 0, (hipStream_t)stream);
}
