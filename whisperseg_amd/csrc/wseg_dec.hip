// Decoder-step kernels for gfx950: token embedding, KV-cache attention (self: per-row gather through
// the beam ancestry table, so a beam reorder never copies the cache; cross: the beams of a window share
// one pass over its 500 encoder keys), and HF-exact greedy / beam-search bookkeeping on the device.
//
// Beam semantics follow HF generation/utils.py:3208-3510 (_beam_search) and helpers :3008-3206 literally,
// including the float32 "+ -1e9" masking arithmetic; see oracle/whisper_ref.py for the CPU restatement.
// These kernels are HBM/latency-bound integer + VALU work (no MFMA): reads are 128-byte rows.
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "wseg_dec.h"

namespace wseg {

// Measurement builds (python -m whisperseg_amd.build --stamps N; never the product library): decode kernel N records the
// 100-MHz s_memrealtime counter at its phase boundaries, one workgroup's thread 0, into g_stamps; tools/stamps.py reads them.
#ifdef WSEG_STAMPS
__device__ unsigned long long g_stamps[32];
#define WSEG_STAMP(K, I) do { if (WSEG_STAMPS == (K) && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g_stamps[I] = wall_clock64(); } while (0)
#else
#define WSEG_STAMP(K, I) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
__global__ void decode_reset_kernel(DecodeState st) {
  for (int w = blockIdx.x * 256 + threadIdx.x; w < st.W; w += gridDim.x * 256) { st.pos[w] = 0; st.done[w] = 1; st.win[w] = -1; st.unsat[w] = 0; st.wmax[w] = st.max_length; }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < st.W * st.nb; i += gridDim.x * 256) st.tokens_in[i] = st.prompt[0];
}

// One workgroup per admitted slot: fresh sequences / scores / ancestry, not done; position pf_np (the prompt positions before it were
// run by the admission pass, run_prompt_pass: their K / V are beam 0's cache rows for every beam).
__global__ __launch_bounds__(256) void decode_admit_kernel(DecodeState st, const int* __restrict__ slots, const int* __restrict__ wins, int pf_np, int pos0) {
  const int w = slots[blockIdx.x];
  const int nb = st.nb, L = st.L;
  for (int i = threadIdx.x; i < nb * L; i += 256) {
    const int p = i % L, j = i / L;
    const int v = p < st.P ? st.prompt[p] : st.pad;
    st.run_seq[(size_t)w * nb * L + i] = v;
    st.fin_seq[(size_t)w * nb * L + i] = v;
    st.anc[(size_t)w * nb * L + i] = (unsigned char)(p < pf_np ? 0 : j);
  }
  if (threadIdx.x < nb) {
    const int j = threadIdx.x, i = w * nb + j;
    st.run_score[i] = j == 0 ? 0.0f : -1.0e9f;
    st.fin_score[i] = -1.0e9f;
    st.fin_flag[i] = 0;
    st.fin_len[i] = 0;
    st.tokens_in[i] = st.prompt[pos0];
  }
  if (threadIdx.x == 0) {
    const int win = wins[blockIdx.x];
    int cap = st.max_length;
    if (st.win_max_length) cap = max(st.P + 1, min(cap, st.win_max_length[win]));
    st.unsat[w] = 1; st.pos[w] = pos0; st.win[w] = win; st.wmax[w] = cap; st.done[w] = 0;
  }
}

__global__ void kv_assign_kernel(int* __restrict__ kv_pt, const int* __restrict__ pairs, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) kv_pt[pairs[2 * i]] = pairs[2 * i + 1];
}

__global__ void decode_abort_kernel(DecodeState st, const int* __restrict__ slots, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < n) { const int w = slots[i]; st.done[w] = 1; st.win[w] = -1; st.unsat[w] = 0; }
}

__global__ void suppress_mask_kernel(unsigned char* mask, int V, const int* sup, int n_sup, const int* bsup, int n_bsup) {
  // single block: clear, then set bits
  for (int i = threadIdx.x; i < V; i += blockDim.x) mask[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_sup; i += blockDim.x) { const int t = sup[i]; if (t >= 0 && t < V) atomicOr((unsigned int*)(mask + (t & ~3)), 1u << (8 * (t & 3))); }
  __syncthreads();
  for (int i = threadIdx.x; i < n_bsup; i += blockDim.x) { const int t = bsup[i]; if (t >= 0 && t < V) atomicOr((unsigned int*)(mask + (t & ~3)), 2u << (8 * (t & 3))); }
}

// T: mode tag.  The token embedding is the LM head's weight matrix (a GEMM operand: hi | lo rows in the split-precision modes,
// read back as hi + lo); the positional table is a plain parameter.
template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(DecodeState st, const void* __restrict__ tok_emb, const typename IO<T>::P* __restrict__ pos_emb,
                                                    float* __restrict__ x, int d) {      // x: the fp32 residual stream
  typedef typename IO<T>::P PT;
  const int r = blockIdx.x;
  const int tok = st.tokens_in[r], pos = st.pos[r / st.nb];
  for (int c = threadIdx.x; c < d; c += 256)
    x[(size_t)r * d + c] = Op<T>::ld1(tok_emb, (size_t)tok, d, c) + El<PT>::ld(pos_emb + (size_t)pos * d + c);
}

// Prompt pass: row i * np + pp = admitted window i at prompt position pp.
template <typename T>
__global__ __launch_bounds__(256) void prompt_embed_kernel(DecodeState st, int np, const void* __restrict__ tok_emb,
                                                           const typename IO<T>::P* __restrict__ pos_emb, float* __restrict__ x, int d) {
  typedef typename IO<T>::P PT;
  const int r = blockIdx.x, pp = r % np;
  const int tok = st.prompt[pp];
  for (int c = threadIdx.x; c < d; c += 256)
    x[(size_t)r * d + c] = Op<T>::ld1(tok_emb, (size_t)tok, d, c) + El<PT>::ld(pos_emb + (size_t)pp * d + c);
}

// ------------------------------------------------------------------------------------------------
// Self-attention, one wave per (row, head).  8 lanes cover one 128-byte K/V row, so one wave instruction
// gathers 8 cache rows (each found through the ancestry table) and 4 instructions are in flight per lane.
// ------------------------------------------------------------------------------------------------
// streaming variant: cross-attention K/V (82 MB per window per step) are read exactly once per step by one workgroup, far more
// than L2 + Infinity Cache hold, so the loads are marked non-temporal.
template <typename T> __device__ __forceinline__ void load8_nt(const T* p, float v[8]) {      // 16-bit element types
  typedef unsigned int nt_u4 __attribute__((ext_vector_type(4)));
  const nt_u4 t = __builtin_nontemporal_load((const nt_u4*)p);
  unpack8<T>(make_uint4(t[0], t[1], t[2], t[3]), v);
}
template <> __device__ __forceinline__ void load8_nt<float>(const float* p, float v[8]) {
  typedef float nt_f4 __attribute__((ext_vector_type(4)));
  const nt_f4 a = __builtin_nontemporal_load((const nt_f4*)p), b = __builtin_nontemporal_load((const nt_f4*)p + 1);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}
template <typename T> __device__ __forceinline__ void load8(const T* p, float v[8]) { unpack8<T>(*(const uint4*)p, v); }
template <> __device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
  const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// sum_z part[z][row][col] + bias[col] in the fixed order z = 0, 1, ... (deterministic), ONE value per thread with every split's load
// issued before the first add; the attention kernels hand the results round through LDS.  (Until round 3 every lane reduced its own
// 8-dim slice, two splits at a time: with 10 splits five DEPENDENT round trips to partials the previous kernel left in memory, per
// call — at 8 slots the four per-beam calls of the cross-attention were most of its 23.5 us.)
template <typename T>
__device__ __forceinline__ float reduce1(const PartialInfo& pi, int row, int col, const T* bias) {
  const float* pp = pi.part + (size_t)row * pi.n + col;
  const size_t zs = (size_t)pi.m_pad * pi.n;
  float t[16];                                           // plan_skinny: at most 16 splits
#pragma unroll
  for (int z = 0; z < 16; ++z) t[z] = pp[(size_t)min(z, pi.splits - 1) * zs];
  float v = 0.f;
#pragma unroll
  for (int z = 0; z < 16; ++z) { if (z < pi.splits) v += t[z]; }
  return bias ? v + El<T>::ld(bias + col) : v;
}

template <typename T> __device__ __forceinline__ void store8(T* p, const float v[8]) { *(uint4*)p = pack8<T>(v); }
template <> __device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
  ((float4*)p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  ((float4*)p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// Sum over the 8 lanes that share a K/V row (lanes 8k .. 8k+7), result in all of them, with DPP adds (no LDS traffic:
// __shfl_xor compiles to ds_bpermute_b32 + s_waitcnt).  Same pairing as xor 1, 2, 4: bit-equal.
__device__ __forceinline__ float row8_sum(float a) {
  a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x141, 0xF, 0xF, true));   // row_half_mirror
  return a;
}

// A K / V cache row slice as loaded (16 bytes of 16-bit elements, or 8 floats), unpacked where it is used: at 1 024 slots the kernel is
// 81 920 single-wave workgroups of dependent loads, so what counts is how many of them a SIMD holds — rows kept as floats cost 77
// registers (6 waves), kept raw 64 (8 waves).
template <typename T> struct Row8 {
  uint4 r;
  __device__ __forceinline__ void ld(const T* p) { r = *(const uint4*)p; }
  __device__ __forceinline__ void get(float v[8]) const { unpack8<T>(r, v); }
};
template <> struct Row8<float> {
  float4 a, b;
  __device__ __forceinline__ void ld(const float* p) { a = ((const float4*)p)[0]; b = ((const float4*)p)[1]; }
  __device__ __forceinline__ void get(float v[8]) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
};

// T: storage type of q / the K / V cache / the bias; TO: tag of the output (the o-proj GEMM's operand: T, or X3<HT> with T = float)
template <typename T, typename TO>
__global__ __launch_bounds__(64, 8) void dec_self_attn_kernel(DecodeState st, const T* __restrict__ q, T* __restrict__ kc,
                                                           T* __restrict__ vc, void* __restrict__ out, int H, int d,
                                                           PartialInfo pi, const T* __restrict__ qkv_bias, float scale) {
  __shared__ float sp[512];
  __shared__ int srow[512];
  __shared__ float snk[64], snv[64];        // this step's own key / value (fused-reduction path)
  __shared__ float sq[64];                  // ... and query
  const int lane = threadIdx.x, sub = lane & 7, rowl = lane >> 3;
  const int r = blockIdx.x / H, h = blockIdx.x - r * H;
  WSEG_STAMP(1, 0);
  const int w = r / st.nb;
  const int L = st.L;
  // ONE round trip for everything that depends on nothing: the idle flag, the position, the first 64 ancestry bytes and (unfused
  // path) the query.  A wave of this kernel is a chain of dependent loads at full occupancy (81 920 single-wave workgroups at
  // 1 024 slots, ten rounds of ~7 us): as separate steps the flag, the position and the ancestry were three links of it.
  const unsigned char* anc = st.anc + (size_t)r * L;
  const int* pt = st.kv_pt + (size_t)w * st.npg;
  const int idle = st.done[w];
  const int pos = st.pos[w];
  int anc0 = anc[min(lane, L - 1)];
  int pg0 = pt[min(lane, L - 1) / KV_PAGE];       // pool unit of position `lane` (same round trip: depends on nothing)
  float qv[8];
  qv[0] = 0.f;
  const bool fused = pi.part != nullptr;
  if (!fused) load8<T>(q + (size_t)r * d + h * 64 + sub * 8, qv);
  asm volatile("" : "+v"(anc0), "+v"(pg0), "+v"(qv[0]));     // keeps the loads above the branch (the compiler would sink them to their uses)
  if (idle) return;                               // idle slot: nothing to append
  WSEG_STAMP(1, 1);
  const int n = pos + 1;                          // keys 0 .. pos (the current token's K/V were just appended)
  if (fused) {
    // finish the split-K reduction of q | k | v for this (row, head): lane e owns dim e of each (all 3 x splits loads in flight
    // together, reduce1); values are rounded to the storage type exactly as the unfused epilogue would have stored and
    // re-loaded them.  The query slices are read back from LDS behind the barrier below.
    const float q1 = reduce1<T>(pi, r, h * 64 + lane, qkv_bias);
    const float k1 = El<T>::rnd(reduce1<T>(pi, r, d + h * 64 + lane, qkv_bias));
    const float v1 = El<T>::rnd(reduce1<T>(pi, r, 2 * d + h * 64 + lane, qkv_bias));
    sq[lane] = El<T>::rnd(q1 * scale);
    // this step's own page: lane n - 1 holds its unit when n <= 64 (no further round trip)
    const int own_unit = n <= 64 ? __builtin_amdgcn_readlane(pg0, __builtin_amdgcn_readfirstlane(n - 1)) : pt[(n - 1) / KV_PAGE];
    const size_t at = ((((size_t)own_unit * st.nb + (r - w * st.nb)) * H + h) * KV_PAGE + ((n - 1) % KV_PAGE)) * 64 + lane;
    El<T>::st(kc + at, k1);
    El<T>::st(vc + at, v1);
    snk[lane] = k1;
    snv[lane] = v1;
  }
  WSEG_STAMP(1, 2);                                     // q | k | v reduced (fused path), cache rows appended
  // srow[t]: index of the 64-element cache row of position t = ((unit nb + beam) H + h) KV_PAGE + t % KV_PAGE
  for (int t = lane; t < n; t += 64) {
    const int beam = t == n - 1 ? (r - w * st.nb) : (t < 64 ? anc0 : (int)anc[t]);
    const int unit = t < 64 ? pg0 : pt[t / KV_PAGE];
    srow[t] = ((unit * st.nb + beam) * H + h) * KV_PAGE + (t % KV_PAGE);
  }
  __syncthreads();
  WSEG_STAMP(1, 3);
  if (fused) {
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = sq[sub * 8 + e];
  }
  // The V rows of the first 32 positions are requested together with the K rows (both depend only on the ancestry
  // table): a wave is one chain of dependent HBM round trips, and this removes one of them for sequences <= 32.
  constexpr bool kPrefetchV = sizeof(T) == 2;          // fp32 rows: the prefetch would cost the 8th wave per SIMD (32 more registers)
  Row8<T> vfirst[4];
  if constexpr (kPrefetchV) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tl = min(u * 8 + rowl, fused ? max(n - 2, 0) : n - 1);
      vfirst[u].ld(vc + (size_t)srow[tl] * 64 + sub * 8);
    }
  }
  for (int t0 = 0; t0 < n; t0 += 32) {
    Row8<T> kr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // unconditional clamped loads (see the cross-attention kernel); the fused step's own key is patched in after
      const int tl = min(t0 + u * 8 + rowl, fused ? max(n - 2, 0) : n - 1);
      kr[u].ld(kc + (size_t)srow[tl] * 64 + sub * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * 8 + rowl;
      float kv[8];
      kr[u].get(kv);
      if (fused && t == n - 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) kv[e] = snk[sub * 8 + e];
      }
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(qv[e], kv[e], s);
      s = row8_sum(s);
      if (sub == 0 && t < n) sp[t] = s;
    }
  }
  WSEG_STAMP(1, 4);                                     // scores
  __syncthreads();
  float mx = -3.0e38f;
  for (int t = lane; t < n; t += 64) mx = fmaxf(mx, sp[t]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int t = lane; t < n; t += 64) { const float p = expf(sp[t] - mx); sp[t] = p; sum += p; }
  sum = wave_sum(sum);
  __syncthreads();
  WSEG_STAMP(1, 5);                                     // softmax
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int t0 = 0; t0 < n; t0 += 32) {
    if (t0 > 0 || !kPrefetchV) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tl = min(t0 + u * 8 + rowl, fused ? max(n - 2, 0) : n - 1);
        vfirst[u].ld(vc + (size_t)srow[tl] * 64 + sub * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = t0 + u * 8 + rowl;
      float vv[8];
      vfirst[u].get(vv);
      if (fused && t == n - 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) vv[e] = snv[sub * 8 + e];
      }
      const float p = t < n ? sp[t] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, vv[e], acc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float a = acc[e];
    a += lane_xor<8>(a);
    a += lane_xor<16>(a);
    a += lane_xor<32>(a);
    acc[e] = a;
  }
  WSEG_STAMP(1, 6);                                     // P V + lane reduction
  if (rowl == 0) {
    const float inv = 1.0f / sum;
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = acc[e] * inv;
    op_st8<TO>(out, (size_t)r, d, h * 64 + sub * 8, o8);
  }
  WSEG_STAMP(1, 7);
}

// Prompt pass (split-precision modes: fp32 cache rows): causal self-attention over the first np <= 4 prompt positions of the windows
// admitted together, one wave per (window, head), lane e = dim e.  Row i * np + pp of q | k | v (split-K partials, or the fp32 rows of
// an un-split GEMM without bias) is window i at position pp; K / V go to beam 0's rows of the slot's first page — the admission
// kernel points every beam's ancestry of these positions there (the forced prompt is the same for all beams).
template <typename TO>
__global__ __launch_bounds__(64) void prompt_self_attn_kernel(const float* __restrict__ qkv, PartialInfo pi, const float* __restrict__ bias,
                                                               float* __restrict__ kc, float* __restrict__ vc, const int* __restrict__ kv_pt, int npg,
                                                               const int* __restrict__ slots, int np, int nb, int H, int d,
                                                               void* __restrict__ out, float scale) {
  __shared__ float so[4][64];
  const int lane = threadIdx.x;
  const int i = blockIdx.x / H, h = blockIdx.x - i * H;
  const int unit = kv_pt[(size_t)slots[i] * npg];
  float q[4], k[4], v[4];
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {
    q[pp] = k[pp] = v[pp] = 0.f;
    if (pp < np) {
      const int row = i * np + pp, col = h * 64 + lane;
      if (pi.part != nullptr) {
        q[pp] = reduce1<float>(pi, row, col, bias);
        k[pp] = reduce1<float>(pi, row, d + col, bias);
        v[pp] = reduce1<float>(pi, row, 2 * d + col, bias);
      } else {
        const float* b = qkv + (size_t)row * 3 * d + col;
        q[pp] = b[0] + bias[col]; k[pp] = b[d] + bias[d + col]; v[pp] = b[2 * d] + bias[2 * d + col];
      }
      q[pp] *= scale;
      const size_t at = ((((size_t)unit * nb) * H + h) * KV_PAGE + pp) * 64 + lane;
      kc[at] = k[pp];
      vc[at] = v[pp];
    }
  }
  static_assert(KV_PAGE >= 4, "the prompt pass writes into the first page");
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {
    if (pp < np) {
      float sc[4], mx = -3.0e38f;
#pragma unroll
      for (int t = 0; t <= pp; ++t) { sc[t] = wave_sum(q[pp] * k[t]); mx = fmaxf(mx, sc[t]); }
      float sum = 0.f, o = 0.f;
#pragma unroll
      for (int t = 0; t <= pp; ++t) { const float pr = expf(sc[t] - mx); sum += pr; o = fmaf(pr, v[t], o); }
      so[pp][lane] = o / sum;
    }
  }
  __syncthreads();
  if (lane < np * 8) {      // lane (pp, e8): 8 consecutive columns of one row — whole quads of lanes share a 32-column block (M6 rows)
    const int pp = lane >> 3, e0 = (lane & 7) * 8;
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = so[pp][e0 + e];
    op_st8<TO>(out, (size_t)(i * np + pp), d, h * 64 + e0, o8);
  }
}

// ------------------------------------------------------------------------------------------------
// Cross-attention, one workgroup per (window, head); all beams of the window in one pass over K and V.
// HBM-bound (128 KiB of K/V per workgroup).  8 lanes cover one 128-byte K/V row (16 B each), so a wave
// reads 8 consecutive rows = 1 KiB fully coalesced per instruction and 4 rows are in flight per lane.
// ------------------------------------------------------------------------------------------------
// (NB = 8, beams 5..8: 64 query registers per lane — two workgroups per CU instead of four; under the 128-register cap that
// instantiation spilled 350 bytes per lane)
template <typename T, typename TO, int NB>
__global__ __launch_bounds__(256, NB > 4 ? 2 : 4) void dec_cross_attn_kernel(DecodeState st, const T* __restrict__ q, const T* __restrict__ ck,
                                                             const T* __restrict__ cv, void* __restrict__ out, int H, int Tk, int d,
                                                             PartialInfo pi, const T* __restrict__ q_bias, float scale) {
  __shared__ float sc[NB][512];
  __shared__ float red[4][NB][64];
  __shared__ float sinv[NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = blockIdx.x / H, h = blockIdx.x - w * H;
  if (st.done[w]) return;                              // idle slot: its 128 KiB of K/V are not streamed
  const int nb = st.nb;
  const int sub = lane & 7, rowl = lane >> 3;          // 8 lanes per row, 8 rows per wave-instruction
  const T* Kb = ck + ((size_t)w * H + h) * Tk * 64;
  const T* Vb = cv + ((size_t)w * H + h) * Tk * 64;
  // this lane's 8-dim slice of every beam's (pre-scaled) query
  float qv[NB][8];
  if (pi.part != nullptr) {
    __shared__ float sq[NB][64];                        // thread (j, e) finishes dim e of beam j (reduce1), slices come back from LDS
    // (NB = 8: 512 (beam, dim) pairs for 256 threads — r05: beams 4..7 used to read their queries from uninitialised LDS whenever the
    // query arrived as split-K partials, i.e. in every mode but f32; found by the first-logit check of tests/test_model_gpu.py)
    for (int i = tid; i < NB * 64; i += 256) sq[i >> 6][i & 63] = El<T>::rnd(reduce1<T>(pi, w * nb + min(i >> 6, nb - 1), h * 64 + (i & 63), q_bias) * scale);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[j][e] = sq[j][sub * 8 + e];
  } else {
#pragma unroll
    for (int j = 0; j < NB; ++j) load8<T>(q + (size_t)(w * nb + min(j, nb - 1)) * d + h * 64 + sub * 8, qv[j]);
  }
  // scores: rows t = it*32 + wave*8 + rowl
  constexpr int U = 4;                                  // K/V rows in flight per lane (8 costs occupancy: measured 1.6x slower)
  for (int t0 = 0; t0 < Tk; t0 += 32 * U) {
    float kv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // unconditional (clamped) loads: a branch around each load makes the compiler wait vmcnt(0) per load and
      // serialises the U rows that are meant to be in flight together; out-of-range rows are discarded below
      const int t = min(t0 + u * 32 + wave * 8 + rowl, Tk - 1);
      load8_nt<T>(Kb + (size_t)t * 64 + sub * 8, kv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * 32 + wave * 8 + rowl;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qv[j][e], kv[u][e], s);
        s = row8_sum(s);
        if (sub == 0 && t < Tk && j < nb) sc[j][t] = s;
      }
    }
  }
  __syncthreads();
  for (int j = wave; j < nb; j += 4) {
    float mx = -3.0e38f;
    for (int t = lane; t < Tk; t += 64) mx = fmaxf(mx, sc[j][t]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < Tk; t += 64) { const float p = expf(sc[j][t] - mx); sc[j][t] = p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) sinv[j] = 1.0f / sum;
  }
  __syncthreads();
  float acc[NB][8];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
  for (int t0 = 0; t0 < Tk; t0 += 32 * U) {
    float vv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = min(t0 + u * 32 + wave * 8 + rowl, Tk - 1);
      load8_nt<T>(Vb + (size_t)t * 64 + sub * 8, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * 32 + wave * 8 + rowl;
      const bool ok = t < Tk;
      const int tc = ok ? t : Tk - 1;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const float p = (ok && j < nb) ? sc[j][tc] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = fmaf(p, vv[u][e], acc[j][e]);
      }
    }
  }
  // reduce over the 8 row-lanes of the wave (lanes with equal `sub`), then over the 4 waves through LDS
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = acc[j][e];
      a += lane_xor<8>(a);
      a += lane_xor<16>(a);
      a += lane_xor<32>(a);
      acc[j][e] = a;
    }
  if (rowl == 0) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave][j][sub * 8 + e] = acc[j][e];
  }
  __syncthreads();
  for (int i = tid; i < nb * 64; i += 256) {
    const int j = i >> 6, e = i & 63;
    const float o = ((red[0][j][e] + red[1][j][e]) + red[2][j][e]) + red[3][j][e];
    Op<TO>::st1(out, (size_t)(w * nb + j), d, h * 64 + e, o * sinv[j]);
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 cross-attention with packed arithmetic — the SAME arithmetic in the same order as the kernel above (bit-equal
// results), at about half the VALU instructions.  The kernel above spends ~120 VALU instructions per 8 K/V rows (8 FMAs
// per beam and row slice, lane reductions through ds_bpermute) and that time (~65 us of a 145-160-us launch at 256
// windows) does not hide under the loads: a plain streaming probe (tools/probes/stream_probe.hip) reads HBM at
// 6.9-7.1 TB/s on this part, the kernel above at 4.1-4.5.  Here: two beams per v_pk_fma_f32 in the score chain and in
// the probability-weighted sum, the 8-lane row sums by DPP adds (quad_perm, quad_perm, row_half_mirror: no LDS), one
// predicated score store per row, rows kept raw (4 VGPRs) until used so that 8 per lane are in flight.
// (v_dot2c_f32_bf16 scores are 8 instructions per row shorter still, but sum in another order — one boundary of the
// tiny-model parity test moved by two mel frames — so the FMA chain stays.)
// ------------------------------------------------------------------------------------------------
template <typename HT, int NB>
__global__ __launch_bounds__(256, 4) void dec_cross_attn_pk_kernel(DecodeState st, const HT* __restrict__ q,
                                                                   const HT* __restrict__ ck, const HT* __restrict__ cv,
                                                                   HT* __restrict__ out, int H, int Tk, int d, PartialInfo pi,
                                                                   const HT* __restrict__ q_bias, float scale) {
  typedef unsigned int raw16 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int U = 8;                                  // K/V rows in flight per lane
  __shared__ float sc[NB][512];
  __shared__ float red[4][NB][64];
  __shared__ float sinv[NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WSEG_STAMP(2, 0);
  const int w = blockIdx.x / H, h = blockIdx.x - w * H;
  if (st.done[w]) return;                              // idle slot: its 128 KiB of K/V are not streamed
  WSEG_STAMP(2, 1);
  const int nb = st.nb;
  const int sub = lane & 7, rowl = lane >> 3;
  const HT* Kb = ck + ((size_t)w * H + h) * Tk * 64;
  const HT* Vb = cv + ((size_t)w * H + h) * Tk * 64;
  // this lane's 8-dim slice of every beam's (pre-scaled) query, fp32, beams paired for v_pk_fma_f32: qq[e][j2] holds
  // dim e of beams 2*j2 and 2*j2+1
  // rows t0 + u*32 + wave*8 + rowl (clamped: out-of-range rows are discarded where they are used), 16 bytes per lane
  auto load_rows = [&](const HT* base, int t0, raw16 (&r)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = min(t0 + u * 32 + wave * 8 + rowl, Tk - 1);
      r[u] = __builtin_nontemporal_load((const raw16*)(base + (size_t)t * 64 + sub * 8));
    }
  };
  constexpr int NP = (NB + 1) / 2;
  f2 qq[8][NP];
  __shared__ float sq[NB][64];                          // thread (j, e) finishes dim e of beam j (reduce1), slices come back from LDS
  if (pi.part != nullptr) {
    if (tid < NB * 64) sq[tid >> 6][tid & 63] = El<HT>::rnd(reduce1<HT>(pi, w * nb + min(tid >> 6, nb - 1), h * 64 + (tid & 63), q_bias) * scale);
    __syncthreads();
  }
  WSEG_STAMP(2, 2);                                 // query reduced
  {
    float qv[8];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (pi.part != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = sq[j][sub * 8 + e];
      } else {
        load8<HT>(q + (size_t)(w * nb + min(j, nb - 1)) * d + h * 64 + sub * 8, qv);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) qq[e][j >> 1][j & 1] = qv[e];
    }
    if constexpr (NB == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) qq[e][0][1] = 0.f;
    }
  }
  // IEEE-half mode: the query is exactly representable in half (it was rounded to the storage type above), so the scores are
  // taken with v_dot2c_f32_f16 straight on the packed K words — half the VALU work of unpack + FMA, fp32 accumulation.  (The
  // bf16 path keeps the FMA chain: it is pinned bit-equal to dec_cross_attn_kernel.)
  constexpr bool kDot2 = std::is_same<HT, f16_t>::value;
  typedef _Float16 hh2 __attribute__((ext_vector_type(2)));
  hh2 qh[NB][4];
  if constexpr (kDot2) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2)
        qh[j][e2] = __builtin_bit_cast(hh2, H16<f16_t>::pack(qq[2 * e2][j >> 1][j & 1], qq[2 * e2 + 1][j >> 1][j & 1]));
  }
  for (int t0 = 0; t0 < Tk; t0 += 32 * U) {
    raw16 kr[U];
    load_rows(Kb, t0, kr);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * 32 + wave * 8 + rowl;
      // scores: the fp32 FMA chain over the 8 dims of the kernel above (bit-equal), two beams per v_pk_fma_f32
      float a[NB];
      if constexpr (kDot2) {
#pragma unroll
        for (int j = 0; j < NB; ++j) a[j] = 0.f;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const unsigned kw = kr[u][e2];        // (a __builtin_bit_cast straight on the vector element picks the wrong lane)
#pragma unroll
          for (int j = 0; j < NB; ++j) a[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(hh2, kw), qh[j][e2], a[j], false);
        }
      } else {
        float kv[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { kv[2 * e] = H16<HT>::lo(kr[u][e]); kv[2 * e + 1] = H16<HT>::hi(kr[u][e]); }
        f2 a2[NP];
#pragma unroll
        for (int j2 = 0; j2 < NP; ++j2) a2[j2] = (f2){0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const f2 kk = {kv[e], kv[e]};
#pragma unroll
          for (int j2 = 0; j2 < NP; ++j2) a2[j2] = __builtin_elementwise_fma(qq[e][j2], kk, a2[j2]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) a[j] = a2[j >> 1][j & 1];
      }
      // the three DPP steps beam-interleaved: a DPP read needs wait states after the write of its source
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[j]), 0xB1, 0xF, 0xF, true));
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[j]), 0x4E, 0xF, 0xF, true));
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[j]), 0x141, 0xF, 0xF, true));
      // lane `sub` of the row stores beam `sub`: one predicated store per row instead of one per beam
      float mine = a[0];
#pragma unroll
      for (int j = 1; j < NB; ++j) mine = sub == j ? a[j] : mine;
      if (sub < nb && t < Tk) sc[sub][t] = mine;
    }
  }
  WSEG_STAMP(2, 3);                                 // scores
  __syncthreads();
  WSEG_STAMP(2, 4);
  for (int j = wave; j < nb; j += 4) {
    float mx = -3.0e38f;
    for (int t = lane; t < Tk; t += 64) mx = fmaxf(mx, sc[j][t]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < Tk; t += 64) { const float p = expf(sc[j][t] - mx); sc[j][t] = p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) sinv[j] = 1.0f / sum;
  }
  __syncthreads();
  WSEG_STAMP(2, 5);                                 // softmax
  static_assert(NB == 1 || NB == 2 || NB == 4, "beam tiles");
  const unsigned sc_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&sc[0][0];
  f2 acc[NB][4];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[j][e] = (f2){0.f, 0.f};
  for (int t0 = 0; t0 < Tk; t0 += 32 * U) {
    raw16 vr[U];
    load_rows(Vb, t0, vr);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * 32 + wave * 8 + rowl;
      const bool ok = t < Tk;
      const int tc = ok ? t : Tk - 1;
      f2 vv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) vv[e] = (f2){H16<HT>::lo(vr[u][e]), H16<HT>::hi(vr[u][e])};
      // probabilities of this row through opaque ds_reads: as C++ loads the compiler gathers all U x NB of them in front
      // of the loop and spills the rows that are in flight (reading the next row's ahead of time costs more registers
      // than it hides: measured slower).  Early-clobber outputs: a result register must not be the address register —
      // other waves issue between these instructions, so an earlier read can land before a later one is issued.
      float pr[NB];
      const unsigned pa = sc_base + (unsigned)tc * 4u;
      if constexpr (NB == 1) asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pr[0]) : "v"(pa) : "memory");
      else if constexpr (NB == 2)
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:2048\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pr[0]), "=&v"(pr[1]) : "v"(pa) : "memory");
      else
        asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:2048\n\tds_read_b32 %2, %4 offset:4096\n\tds_read_b32 %3, %4 offset:6144\n\t"
                     "s_waitcnt lgkmcnt(0)" : "=&v"(pr[0]), "=&v"(pr[1]), "=&v"(pr[2]), "=&v"(pr[3]) : "v"(pa) : "memory");
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const float p = (ok && j < nb) ? pr[j] : 0.f;
        const f2 pp = {p, p};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = __builtin_elementwise_fma(pp, vv[e], acc[j][e]);
      }
    }
  }
  // reduce over the 8 row-lanes of the wave (lanes with equal `sub`), then over the 4 waves through LDS
  WSEG_STAMP(2, 6);                                 // P V
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = acc[j][e >> 1][e & 1];
      a += lane_xor<8>(a);
      a += lane_xor<16>(a);
      a += lane_xor<32>(a);
      if (rowl == 0) red[wave][j][sub * 8 + e] = a;
    }
  __syncthreads();
  for (int i = tid; i < nb * 64; i += 256) {
    const int j = i >> 6, e = i & 63;
    const float o = ((red[0][j][e] + red[1][j][e]) + red[2][j][e]) + red[3][j][e];
    El<HT>::st(out + (size_t)(w * nb + j) * d + h * 64 + e, o * sinv[j]);
  }
  WSEG_STAMP(2, 7);
}

// ------------------------------------------------------------------------------------------------
// Split-precision modes: cross-attention over 24-bit K / V (EpiParams::kv24: per (slot, head) a [Tk][64] plane of the fp32
// words' top halves, then a [Tk][64] plane of their third bytes; 192 instead of 256 bytes per row pair of an HBM-bound stream).
// fp32 query and arithmetic; same structure as the packed 16-bit kernel above: 8 lanes per row, 8 raw rows per lane in flight
// (16 + 8 bytes each), one v_perm_b32 per element to rebuild the fp32 word, two beams per v_pk_fma_f32, DPP row sums.
// BFP (r06, EpiParams::kv24 == 3, the x3 modes' format): the same two planes hold 24-bit two's-complement INTEGERS and a third plane of
// [Tk] fp32 row scales follows (196 bytes per row; st_bfp24_row in wseg_gemm_epi.h): the rebuilt word is 256 q, one v_cvt_f32_i32 per
// element more, and the row scales are applied where a row is one number — the K scale to the finished score and the V scale to the
// probability, both in the softmax pass over LDS — so neither streaming loop carries another register.
// ------------------------------------------------------------------------------------------------
#ifndef WSEG_CA_PREFETCH
#define WSEG_CA_PREFETCH 1      // first V rows requested in front of the softmax pass, row scales parked in LDS (A/B: build --variant nopf -DWSEG_CA_PREFETCH=0)
#endif
#ifndef WSEG_CA_PREFETCH_K
#define WSEG_CA_PREFETCH_K 0    // first K rows in front of the query reduction: rows live across reduce1 spill at 128 registers (r06: 8 rows 8-24 VGPRs, 4 rows 4-31)
#endif
template <typename TO, int NB, bool BFP>
__global__ __launch_bounds__(256, 4) void dec_cross_attn_k24_kernel(DecodeState st, const float* __restrict__ q,
                                                                    const unsigned char* __restrict__ ck, const unsigned char* __restrict__ cv,
                                                                    void* __restrict__ out, int H, int Tk, int d, PartialInfo pi,
                                                                    const float* __restrict__ q_bias, float scale, const int* __restrict__ kv_slot) {
  typedef unsigned int raw16 __attribute__((ext_vector_type(4)));
  typedef unsigned int raw8 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int U = 8;
  __shared__ __attribute__((aligned(16))) float sc[512][NB];      // [position][beam]: one read per row in the V pass
  __shared__ float red[4][NB][64];
  __shared__ float sinv[NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  WSEG_STAMP(3, 0);
  const int w = blockIdx.x / H, h = blockIdx.x - w * H;
  if (st.done[w]) return;                              // idle slot: its 96 KiB of K/V are not streamed
  WSEG_STAMP(3, 1);
  const int nb = st.nb;
  const int sub = lane & 7, rowl = lane >> 3;
  const int ws = kv_slot ? kv_slot[w] : w;              // prompt pass: query rows of admitted window w, K / V of its slot
  constexpr int ROWB = BFP ? 196 : 192;
  const unsigned char* Kb = ck + ((size_t)ws * H + h) * Tk * ROWB;
  const unsigned char* Vb = cv + ((size_t)ws * H + h) * Tk * ROWB;
  const unsigned char* Kl = Kb + (size_t)Tk * 128;
  const unsigned char* Vl = Vb + (size_t)Tk * 128;
  const float* Ks = (const float*)(Kb + (size_t)Tk * 192);      // BFP: row scales
  const float* Vs = (const float*)(Vb + (size_t)Tk * 192);
  // BFP: the row scales wait in LDS for the softmax pass (read from global there, each wave's first access paid an HBM round trip with no
  // stream in flight); visible behind the barrier that closes the score pass
  __shared__ float sks[BFP && WSEG_CA_PREFETCH ? 512 : 1], svs[BFP && WSEG_CA_PREFETCH ? 512 : 1];
  if constexpr (BFP && WSEG_CA_PREFETCH) {
    for (int t = tid; t < Tk; t += 256) { sks[t] = Ks[t]; svs[t] = Vs[t]; }
  }
  constexpr int NP = (NB + 1) / 2;
  f2 qq[8][NP];
  __shared__ float sq[NB][64];                          // thread (j, e) finishes dim e of beam j (reduce1), slices come back from LDS
  raw16 kh[U];
  raw8 kl[U];
  auto load_k = [&](int t0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = min(t0 + u * 32 + wave * 8 + rowl, Tk - 1);      // clamped: out-of-range rows are discarded below
      kh[u] = __builtin_nontemporal_load((const raw16*)(Kb + (size_t)t * 128 + sub * 16));
      kl[u] = __builtin_nontemporal_load((const raw8*)(Kl + (size_t)t * 64 + sub * 8));
    }
  };
  if constexpr (WSEG_CA_PREFETCH_K) load_k(0);          // the stream starts under the query's split-K reduction
  if (pi.part != nullptr) {
    if (tid < NB * 64) sq[tid >> 6][tid & 63] = reduce1<float>(pi, w * nb + min(tid >> 6, nb - 1), h * 64 + (tid & 63), q_bias) * scale;
    __syncthreads();
  }
  WSEG_STAMP(3, 2);                                 // query reduced
  {
    float qv[8];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (pi.part != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = sq[j][sub * 8 + e];
      } else {
        load8<float>(q + (size_t)(w * nb + min(j, nb - 1)) * d + h * 64 + sub * 8, qv);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) qq[e][j >> 1][j & 1] = qv[e];
    }
    if constexpr (NB == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) qq[e][0][1] = 0.f;
    }
  }
  // element e of a row slice: 32-bit word = [top half e][third byte e][0] — an fp32, or (BFP) the integer 256 q
  auto unpack = [](const raw16& hi, const raw8& lo, float v[8]) {
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      const unsigned hw = hi[e2], lw = lo[e2 >> 1];
      const unsigned s0 = (e2 & 1) ? 0x0504020cu : 0x0504000cu, s1 = (e2 & 1) ? 0x0706030cu : 0x0706010cu;
      const unsigned w0 = __builtin_amdgcn_perm(hw, lw, s0), w1 = __builtin_amdgcn_perm(hw, lw, s1);
      v[2 * e2] = BFP ? (float)(int)w0 : __uint_as_float(w0);
      v[2 * e2 + 1] = BFP ? (float)(int)w1 : __uint_as_float(w1);
    }
  };
  for (int t0 = 0; t0 < Tk; t0 += 32 * U) {
    if (!WSEG_CA_PREFETCH_K || t0 > 0) load_k(t0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * 32 + wave * 8 + rowl;
      float kv[8];
      unpack(kh[u], kl[u], kv);
      f2 a2[NP];
#pragma unroll
      for (int j2 = 0; j2 < NP; ++j2) a2[j2] = (f2){0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const f2 kk = {kv[e], kv[e]};
#pragma unroll
        for (int j2 = 0; j2 < NP; ++j2) a2[j2] = __builtin_elementwise_fma(qq[e][j2], kk, a2[j2]);
      }
      float a[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] = a2[j >> 1][j & 1];
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[j]), 0xB1, 0xF, 0xF, true));
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[j]), 0x4E, 0xF, 0xF, true));
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[j]), 0x141, 0xF, 0xF, true));
      float mine = a[0];
#pragma unroll
      for (int j = 1; j < NB; ++j) mine = sub == j ? a[j] : mine;
      if (sub < nb && t < Tk) sc[t][sub] = mine;
    }
  }
  WSEG_STAMP(3, 3);                                 // scores
  // The first V rows are requested HERE, in front of the softmax pass (the K registers are dead: no register is added), so the stream does
  // not stop between the two passes: their HBM round trip runs under the softmax instead of behind it.
  raw16 vh[U];
  raw8 vl[U];
  auto load_v = [&](int t0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = min(t0 + u * 32 + wave * 8 + rowl, Tk - 1);
      vh[u] = __builtin_nontemporal_load((const raw16*)(Vb + (size_t)t * 128 + sub * 16));
      vl[u] = __builtin_nontemporal_load((const raw8*)(Vl + (size_t)t * 64 + sub * 8));
    }
  };
  if constexpr (WSEG_CA_PREFETCH) load_v(0);
  __syncthreads();
  WSEG_STAMP(3, 4);
  for (int j = wave; j < nb; j += 4) {
    float mx = -3.0e38f;
    if constexpr (BFP) {
      for (int t = lane; t < Tk; t += 64) { const float x = sc[t][j] * (WSEG_CA_PREFETCH ? sks[t] : Ks[t]); sc[t][j] = x; mx = fmaxf(mx, x); }
    } else {
      for (int t = lane; t < Tk; t += 64) mx = fmaxf(mx, sc[t][j]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < Tk; t += 64) { const float p = expf(sc[t][j] - mx); sc[t][j] = BFP ? p * (WSEG_CA_PREFETCH ? svs[t] : Vs[t]) : p; sum += p; }
    sum = wave_sum(sum);
    if (lane == 0) sinv[j] = 1.0f / sum;
  }
  __syncthreads();
  WSEG_STAMP(3, 5);                                 // softmax
  static_assert(NB == 1 || NB == 2 || NB == 4, "beam tiles");
  f2 acc[NB][4];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[j][e] = (f2){0.f, 0.f};
  for (int t0 = 0; t0 < Tk; t0 += 32 * U) {
    if (!WSEG_CA_PREFETCH || t0 > 0) load_v(t0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * 32 + wave * 8 + rowl;
      const bool ok = t < Tk;
      const int tc = ok ? t : Tk - 1;
      float vf[8];
      unpack(vh[u], vl[u], vf);
      float pr[NB];
      if constexpr (NB == 4) { const float4 t4 = *(const float4*)&sc[tc][0]; pr[0] = t4.x; pr[1] = t4.y; pr[2] = t4.z; pr[3] = t4.w; }
      else if constexpr (NB == 2) { const float2 t2 = *(const float2*)&sc[tc][0]; pr[0] = t2.x; pr[1] = t2.y; }
      else pr[0] = sc[tc][0];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const float p = (ok && j < nb) ? pr[j] : 0.f;
        const f2 pp = {p, p};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = __builtin_elementwise_fma(pp, (f2){vf[2 * e], vf[2 * e + 1]}, acc[j][e]);
      }
    }
  }
  WSEG_STAMP(3, 6);                                 // P V
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = acc[j][e >> 1][e & 1];
      a += lane_xor<8>(a);
      a += lane_xor<16>(a);
      a += lane_xor<32>(a);
      if (rowl == 0) red[wave][j][sub * 8 + e] = a;
    }
  __syncthreads();
  if (tid < NB * 8) {      // thread (j, e8): 8 consecutive columns of beam j — whole quads of threads share a 32-column block (M6 rows)
    const int j = tid >> 3, e0 = (tid & 7) * 8;
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (((red[0][j][e0 + e] + red[1][j][e0 + e]) + red[2][j][e0 + e]) + red[3][j][e0 + e]) * sinv[min(j, nb - 1)];
    if (j < nb) op_st8<TO>(out, (size_t)(w * nb + j), d, h * 64 + e0, o8);
  }
  WSEG_STAMP(3, 7);
}

// ------------------------------------------------------------------------------------------------
// Split-precision modes, r05: cross-attention over block-floating-point K / V (EpiParams::kv24 == 2: per (slot, head) a [Tk][64] plane
// of int16 followed by [Tk] fp32 powers of two, value = int16 * scale of its row; 132 instead of the 24-bit format's 192 bytes per
// row of an HBM-bound stream).  The structure is the 24-bit kernel's: 8 lanes per row, 8 raw rows per lane in flight (16 + 4 bytes
// each), two beams per v_pk_fma_f32, DPP row sums; an element is one v_cvt_f32_i32 (sign-extended half word), the row's scale
// multiplies the finished score (K) / the probability (V) — powers of two: exact, the sums are those of the dequantised values.
// ------------------------------------------------------------------------------------------------
#ifndef WSEG_BFP_U
#define WSEG_BFP_U 8
#endif
// Occupancy: four workgroups per CU (128 registers).  The 4-beam instantiation fits with 5 instead of 8 V rows per lane in flight and the
// scores as [position][beam] (one 16-byte LDS read per row instead of four addresses); with 8 it needs 168 registers = three per CU.
#ifndef WSEG_BFP_OCC
#define WSEG_BFP_OCC 4
#endif
#ifndef WSEG_BFP_UV
#define WSEG_BFP_UV (WSEG_BFP_OCC == 4 ? 5 : 8)
#endif
template <typename TO, int NB>
__global__ __launch_bounds__(256, WSEG_BFP_OCC) void dec_cross_attn_bfp_kernel(DecodeState st, const float* __restrict__ q,
                                                                    const unsigned char* __restrict__ ck, const unsigned char* __restrict__ cv,
                                                                    void* __restrict__ out, int H, int Tk, int d, PartialInfo pi,
                                                                    const float* __restrict__ q_bias, float scale, const int* __restrict__ kv_slot) {
  typedef unsigned int raw16 __attribute__((ext_vector_type(4)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int U = WSEG_BFP_U;
  __shared__ __attribute__((aligned(16))) float sc[512][NB];      // [position][beam]: one 16-byte read per row in the V pass
  __shared__ float red[4][NB][64];
  __shared__ float sinv[NB];
  __shared__ float sks[512], svs[512];                 // the row scales of this (slot, head): 2 x 2 KB, fetched once with coalesced loads (as
                                                       // 4-byte loads beside the rows they doubled the kernel's VMEM instructions: 5.2 TB/s)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = blockIdx.x / H, h = blockIdx.x - w * H;
  if (st.done[w]) return;                              // idle slot: its 66 KB of K / V are not streamed
  const int nb = st.nb;
  const int sub = lane & 7, rowl = lane >> 3;
  const int ws = kv_slot ? kv_slot[w] : w;              // prompt pass: query rows of admitted window w, K / V of its slot
  const unsigned char* Kb = ck + ((size_t)ws * H + h) * Tk * 132;
  const unsigned char* Vb = cv + ((size_t)ws * H + h) * Tk * 132;
  const float* Ks = (const float*)(Kb + (size_t)Tk * 128);
  const float* Vs = (const float*)(Vb + (size_t)Tk * 128);
  constexpr int NP = (NB + 1) / 2;
  f2 qq[8][NP];
  __shared__ float sq[NB][64];                          // thread (j, e) finishes dim e of beam j (reduce1), slices come back from LDS
  // The first batch of K rows is requested BEFORE the query is assembled (reduce1: a chain of dependent split-K loads) and the first
  // batch of V rows before the softmax: neither depends on what it overtakes, and the kernel's fixed costs (~170 us of a 464-us launch at
  // 1 024 slots by a two-point fit against the 24-bit format) are exactly these serial sections.
  constexpr int UV = NB == 4 ? WSEG_BFP_UV : U;      // rows per lane in flight in the V pass (the 4-beam accumulators leave room for 6 at 128 registers)
  auto ld_rows = [&](const unsigned char* base, int t0, auto& r) __attribute__((always_inline)) {
    constexpr int UU = sizeof(r) / sizeof(r[0]);
#pragma unroll
    for (int u = 0; u < UU; ++u) {
      const int t = min(t0 + u * 32 + wave * 8 + rowl, Tk - 1);      // clamped: out-of-range rows are discarded by their consumers
      r[u] = __builtin_nontemporal_load((const raw16*)(base + (size_t)t * 128 + sub * 16));
    }
  };
#ifndef WSEG_BFP_PREFETCH
#define WSEG_BFP_PREFETCH 1
#endif
  raw16 kq[U];
  if (WSEG_BFP_PREFETCH) ld_rows(Kb, 0, kq);
  for (int i = tid; i < Tk; i += 256) { sks[i] = __builtin_nontemporal_load(Ks + i); svs[i] = __builtin_nontemporal_load(Vs + i); }
  if (pi.part != nullptr) {
    if (tid < NB * 64) sq[tid >> 6][tid & 63] = reduce1<float>(pi, w * nb + min(tid >> 6, nb - 1), h * 64 + (tid & 63), q_bias) * scale;
  }
  __syncthreads();
  {
    float qv[8];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (pi.part != nullptr) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = sq[j][sub * 8 + e];
      } else {
        load8<float>(q + (size_t)(w * nb + min(j, nb - 1)) * d + h * 64 + sub * 8, qv);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) qq[e][j >> 1][j & 1] = qv[e];
    }
    if constexpr (NB == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) qq[e][0][1] = 0.f;
    }
  }
  auto unpack = [](const raw16& x, float v[8]) {
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      v[2 * e2] = (float)(int)(short)(x[e2] & 0xffffu);
      v[2 * e2 + 1] = (float)((int)x[e2] >> 16);
    }
  };
  for (int t0 = 0; t0 < Tk; t0 += 32 * U) {
    if (t0 > 0 || !WSEG_BFP_PREFETCH) ld_rows(Kb, t0, kq);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * 32 + wave * 8 + rowl;
      float kv[8];
      unpack(kq[u], kv);
      f2 a2[NP];
#pragma unroll
      for (int j2 = 0; j2 < NP; ++j2) a2[j2] = (f2){0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const f2 kk = {kv[e], kv[e]};
#pragma unroll
        for (int j2 = 0; j2 < NP; ++j2) a2[j2] = __builtin_elementwise_fma(qq[e][j2], kk, a2[j2]);
      }
      float a[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] = a2[j >> 1][j & 1];
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[j]), 0xB1, 0xF, 0xF, true));
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[j]), 0x4E, 0xF, 0xF, true));
#pragma unroll
      for (int j = 0; j < NB; ++j) a[j] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a[j]), 0x141, 0xF, 0xF, true));
      float mine = a[0];
#pragma unroll
      for (int j = 1; j < NB; ++j) mine = sub == j ? a[j] : mine;
      if (sub < nb && t < Tk) sc[t][sub] = mine * sks[t];
    }
  }
  raw16 vq[UV];
  if (WSEG_BFP_PREFETCH) ld_rows(Vb, 0, vq);
  __syncthreads();
  for (int j = wave; j < nb; j += 4) {
    float mx = -3.0e38f;
    for (int t = lane; t < Tk; t += 64) mx = fmaxf(mx, sc[t][j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < Tk; t += 64) { const float p = expf(sc[t][j] - mx); sc[t][j] = p * svs[t]; sum += p; }      // (V's row scale rides on the probability)
    sum = wave_sum(sum);
    if (lane == 0) sinv[j] = 1.0f / sum;
  }
  __syncthreads();
  static_assert(NB == 1 || NB == 2 || NB == 4, "beam tiles");
  f2 acc[NB][4];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[j][e] = (f2){0.f, 0.f};
  for (int t0 = 0; t0 < Tk; t0 += 32 * UV) {
    if (t0 > 0 || !WSEG_BFP_PREFETCH) ld_rows(Vb, t0, vq);
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      const int t = t0 + u * 32 + wave * 8 + rowl;
      const bool ok = t < Tk;
      const int tc = ok ? t : Tk - 1;
      float vf[8];
      unpack(vq[u], vf);
      float pr[NB];
      if constexpr (NB == 4) { const float4 t4 = *(const float4*)&sc[tc][0]; pr[0] = t4.x; pr[1] = t4.y; pr[2] = t4.z; pr[3] = t4.w; }
      else if constexpr (NB == 2) { const float2 t2 = *(const float2*)&sc[tc][0]; pr[0] = t2.x; pr[1] = t2.y; }
      else pr[0] = sc[tc][0];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const float p = (ok && j < nb) ? pr[j] : 0.f;
        const f2 pp = {p, p};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][e] = __builtin_elementwise_fma(pp, (f2){vf[2 * e], vf[2 * e + 1]}, acc[j][e]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = acc[j][e >> 1][e & 1];
      a += lane_xor<8>(a);
      a += lane_xor<16>(a);
      a += lane_xor<32>(a);
      if (rowl == 0) red[wave][j][sub * 8 + e] = a;
    }
  __syncthreads();
  if (tid < NB * 8) {      // thread (j, e8): 8 consecutive columns of beam j — whole quads of threads share a 32-column block (M6 rows)
    const int j = tid >> 3, e0 = (tid & 7) * 8;
    float o8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o8[e] = (((red[0][j][e0 + e] + red[1][j][e0 + e]) + red[2][j][e0 + e]) + red[3][j][e0 + e]) * sinv[min(j, nb - 1)];
    if (j < nb) op_st8<TO>(out, (size_t)(w * nb + j), d, h * 64 + e0, o8);
  }
}

// ------------------------------------------------------------------------------------------------
// Per-row candidates: log_softmax (fp32) -> suppress -> + running beam score -> top-Kc.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool better(float av, int ai, float bv, int bi) { return av > bv || (av == bv && ai < bi); }

// Stage 1: grid (NSEG, R).  Each workgroup scans one slice of a row's logits: slice max, slice
// sum(exp(x - slice max)) and the slice's top-KC of the processed logits (suppressed ids -> -inf).
// log_softmax is a per-row shift, so ranking raw logits == ranking log-probs; the shift is applied to the
// few surviving candidates in stage 2.
template <int KC>
// list != nullptr (the first generated step of newly admitted windows, run by the admission): grid.y = n_list * nb, state row r = the row of
// beam j of slot list[i], logits row i (one row per window: its beams are copies at that step)
__global__ __launch_bounds__(256) void row_topk_partial_kernel(DecodeState st, const float* __restrict__ logits, int nseg,
                                                               float* __restrict__ part_val, int* __restrict__ part_idx,
                                                               float* __restrict__ part_stat, const int* __restrict__ list) {
  __shared__ float s_red[4];
  __shared__ float s_thr[4];
  __shared__ float s_bv[4];
  __shared__ int s_bi[4];
  __shared__ int s_bt[4];
  __shared__ int s_winner;
  const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int r = blockIdx.y, xr = r;
  if (list) { xr = r / st.nb; r = list[xr] * st.nb + (r - xr * st.nb); }
  if (st.done[r / st.nb]) return;
  const int V = st.V;
  const int per = (V + nseg - 1) / nseg;
  const int lo = seg * per, hi = min(V, lo + per);
  const float* x = logits + (size_t)xr * st.ldv;
  const int cur_len = st.pos[r / st.nb] + 1;
  const unsigned char bits = 1 | ((cur_len == st.P) ? 2 : 0);
  // the slice as [lo, a4) scalar head, [a4, b4) 16-byte groups (rows are 16-byte aligned: ldv % 4 == 0), [b4, hi) tail:
  // a row is 200 KB of fp32, one dword per lane per load left the scan latency-bound (0.66 TB/s at 1024 rows)
  const int a4 = min(hi, (lo + 3) & ~3), b4 = max(a4, hi & ~3);
  // pass 1: the slice maximum of the raw logits (for the log-sum-exp) and, per thread, the maximum of the PROCESSED logits
  // (suppressed ids -> -inf), from which a lower bound on the slice's KC-th best candidate follows
  float mx = -3.0e38f, pm = -INFINITY;
  auto seen = [&](float v, unsigned char sup) { mx = fmaxf(mx, v); if (!(sup & bits)) pm = fmaxf(pm, v); };
  for (int i = lo + tid; i < a4; i += 256) seen(x[i], st.sup_mask[i]);
#pragma unroll 8
  for (int i = a4 + tid * 4; i < b4; i += 1024) {          // unrolled: 8 independent 16-byte loads in flight per lane
    const float4 v = *(const float4*)(x + i);
    const uchar4 m4 = *(const uchar4*)(st.sup_mask + i);
    seen(v.x, m4.x); seen(v.y, m4.y); seen(v.z, m4.z); seen(v.w, m4.w);
  }
  for (int i = b4 + tid; i < hi; i += 256) seen(x[i], st.sup_mask[i]);
  mx = wave_max(mx);
  // KC-th largest of the wave's 64 per-lane maxima: at least KC distinct elements of the slice are >= it, so nothing below it
  // can be among the slice's KC best.  Without this filter the sorted insertion below runs for nearly every element: some
  // lane of the wave beats its own KC-th best at almost every step, and the whole wave pays for it.
  float thr = -INFINITY;
  {
    float cand = pm;
#pragma unroll 1
    for (int k = 0; k < KC; ++k) {
      thr = wave_max(cand);
      const unsigned long long hit = __builtin_amdgcn_ballot_w64(cand == thr);
      if (lane == __ffsll((long long)hit) - 1) cand = -INFINITY;      // drop ONE instance
    }
  }
  if (lane == 0) { s_red[wave] = mx; s_thr[wave] = thr; }
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  thr = fmaxf(fmaxf(s_thr[0], s_thr[1]), fmaxf(s_thr[2], s_thr[3]));      // each wave's bound holds: take the tightest
  __syncthreads();
  float sum = 0.f;
  float tv[KC];
  int ti[KC];
#pragma unroll
  for (int j = 0; j < KC; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
  auto visit = [&](float v, int i, unsigned char sup) {
    sum += expf(v - mx);
    if (sup & bits) v = -INFINITY;
    if (v >= thr && better(v, i, tv[KC - 1], ti[KC - 1])) {
#pragma unroll
      for (int j = KC - 1; j >= 0; --j) {
        const bool gt_prev = (j > 0) ? better(v, i, tv[j > 0 ? j - 1 : 0], ti[j > 0 ? j - 1 : 0]) : false;
        const bool gt_cur = better(v, i, tv[j], ti[j]);
        if (gt_prev) { tv[j] = tv[j - 1 >= 0 ? j - 1 : 0]; ti[j] = ti[j - 1 >= 0 ? j - 1 : 0]; }
        else if (gt_cur) { tv[j] = v; ti[j] = i; }
      }
    }
  };
  for (int i = lo + tid; i < a4; i += 256) visit(x[i], i, st.sup_mask[i]);
#pragma unroll 4
  for (int i = a4 + tid * 4; i < b4; i += 1024) {
    const float4 v = *(const float4*)(x + i);
    const uchar4 m4 = *(const uchar4*)(st.sup_mask + i);      // the mask buffer is 4-byte aligned and padded (align_up(V, 4))
    visit(v.x, i, m4.x);
    visit(v.y, i + 1, m4.y);
    visit(v.z, i + 2, m4.z);
    visit(v.w, i + 3, m4.w);
  }
  for (int i = b4 + tid; i < hi; i += 256) visit(x[i], i, st.sup_mask[i]);
  sum = wave_sum(sum);
  if (lane == 0) s_red[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    part_stat[((size_t)r * nseg + seg) * 2 + 0] = mx;
    part_stat[((size_t)r * nseg + seg) * 2 + 1] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  }
  for (int k = 0; k < KC; ++k) {
    float bv = tv[0];
    int bi = ti[0], bt = tid;
#define WSEG_ARGMAX_STEP(O) { const float ov = lane_xor<O>(bv); const int oi = lane_xor<O>(bi), ot = lane_xor<O>(bt); \
                             if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; bt = ot; } }
    WSEG_ARGMAX_STEP(32) WSEG_ARGMAX_STEP(16) WSEG_ARGMAX_STEP(8) WSEG_ARGMAX_STEP(4) WSEG_ARGMAX_STEP(2) WSEG_ARGMAX_STEP(1)
#undef WSEG_ARGMAX_STEP
    if (lane == 0) { s_bv[wave] = bv; s_bi[wave] = bi; s_bt[wave] = bt; }
    __syncthreads();
    if (tid == 0) {
      float fv = s_bv[0]; int fi = s_bi[0], ft = s_bt[0];
      for (int q = 1; q < 4; ++q) if (better(s_bv[q], s_bi[q], fv, fi)) { fv = s_bv[q]; fi = s_bi[q]; ft = s_bt[q]; }
      part_val[((size_t)r * nseg + seg) * KC + k] = fv;
      part_idx[((size_t)r * nseg + seg) * KC + k] = fi;
      s_winner = ft;
    }
    __syncthreads();
    if (tid == s_winner) {
#pragma unroll
      for (int j = 0; j < KC - 1; ++j) { tv[j] = tv[j + 1]; ti[j] = ti[j + 1]; }
      tv[KC - 1] = -INFINITY; ti[KC - 1] = 0x7fffffff;
    }
    __syncthreads();
  }
}

// Stage 2: one wave per row merges the slices: global max / log-sum-exp, then the row's top-Kc candidates as
// log_softmax(x) (+ -inf for suppressed ids) + running beam score — HF generation/utils.py:3374-3395.
template <int KC>
__global__ __launch_bounds__(64) void row_topk_merge_kernel(DecodeState st, int nseg, const float* __restrict__ part_val,
                                                            const int* __restrict__ part_idx, const float* __restrict__ part_stat,
                                                            const int* __restrict__ list) {
  const int lane = threadIdx.x;
  int r = blockIdx.x;
  if (list) { const int i = r / st.nb; r = list[i] * st.nb + (r - i * st.nb); }
  if (st.done[r / st.nb]) return;
  const bool greedy = st.nb == 1;
  const int Kc = greedy ? st.top_k : 2 * st.nb;          // greedy: 1 candidate; sampling: top_k raw logits
  float mx = -3.0e38f;
  for (int s = lane; s < nseg; s += 64) mx = fmaxf(mx, part_stat[((size_t)r * nseg + s) * 2]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int s = lane; s < nseg; s += 64) sum += part_stat[((size_t)r * nseg + s) * 2 + 1] * expf(part_stat[((size_t)r * nseg + s) * 2] - mx);
  sum = wave_sum(sum);
  const float lse = logf(sum);
  const float base = greedy ? 0.f : st.run_score[r];
  const int n = nseg * KC;                 // <= 16 * 16 = 256 candidates: 4 per lane
  float cv[4];
  int ci[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = lane + u * 64;
    cv[u] = c < n ? part_val[(size_t)r * n + c] : -INFINITY;
    ci[u] = c < n ? part_idx[(size_t)r * n + c] : 0x7fffffff;
  }
  for (int k = 0; k < Kc; ++k) {
    float bv = cv[0]; int bi = ci[0], bu = 0;
#pragma unroll
    for (int u = 1; u < 4; ++u) if (better(cv[u], ci[u], bv, bi)) { bv = cv[u]; bi = ci[u]; bu = u; }
    float wv = bv; int wi = bi, wl = lane;
#define WSEG_ARGMAX_STEP(O) { const float ov = lane_xor<O>(wv); const int oi = lane_xor<O>(wi), ol = lane_xor<O>(wl); \
                             if (better(ov, oi, wv, wi)) { wv = ov; wi = oi; wl = ol; } }
    WSEG_ARGMAX_STEP(32) WSEG_ARGMAX_STEP(16) WSEG_ARGMAX_STEP(8) WSEG_ARGMAX_STEP(4) WSEG_ARGMAX_STEP(2) WSEG_ARGMAX_STEP(1)
#undef WSEG_ARGMAX_STEP
    if (lane == wl) {
#pragma unroll
      for (int u = 0; u < 4; ++u) if (u == bu) { cv[u] = -INFINITY; ci[u] = 0x7fffffff; }
    }
    if (lane == 0) {
      float v = wv;
      if (!greedy) v = ((wv - mx) - lse) + base;     // -inf stays -inf
      st.cand_val[(size_t)r * Kc + k] = v;
      st.cand_tok[(size_t)r * Kc + k] = wi;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Beam bookkeeping: one 64-lane workgroup per window (lane 0 decides, all lanes move sequences).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void beam_step_kernel(DecodeState st, const int* __restrict__ list) {
  __shared__ float c_val[MAX_CAND];
  __shared__ int c_beam[MAX_CAND], c_tok[MAX_CAND], c_hit[MAX_CAND];
  __shared__ float run_val[MAX_CAND], fin_cand[MAX_CAND];
  __shared__ int run_src[MAX_BEAMS], run_tok[MAX_BEAMS], fin_src[MAX_BEAMS];
  __shared__ float n_run_score[MAX_BEAMS], n_fin_score[MAX_BEAMS];
  __shared__ int n_fin_flag[MAX_BEAMS], n_fin_len[MAX_BEAMS];
  __shared__ int head[MAX_BEAMS];
  const int w = list ? list[blockIdx.x] : blockIdx.x, lane = threadIdx.x;
  if (st.done[w]) return;               // idle slot
  const int nb = st.nb, Kc = 2 * nb, L = st.L, P = st.P;
  const int pos = st.pos[w];
  const int cur_len = pos + 1;          // tokens present before this step's choice
  if (cur_len < P) {                    // prompt phase: the next token is forced, nothing else changes
    if (lane < nb) st.tokens_in[w * nb + lane] = st.prompt[cur_len];
    if (lane == 0) st.pos[w] = pos + 1;
    return;
  }
  if (lane == 0) {
    // c. top-Kc continuations over nb*V accumulated log-probs (each row's list is already sorted)
    for (int j = 0; j < nb; ++j) head[j] = 0;
    for (int k = 0; k < Kc; ++k) {
      int bj = -1; float bv = 0.f; int bt = 0;
      for (int j = 0; j < nb; ++j) {
        if (head[j] >= Kc) continue;
        const size_t o = (size_t)(w * nb + j) * Kc + head[j];
        const float v = st.cand_val[o];
        const int t = st.cand_tok[o];
        if (bj < 0 || v > bv) { bj = j; bv = v; bt = t; }   // ties keep the lower flat index (lower beam)
      }
      head[bj]++;
      c_val[k] = bv; c_beam[k] = bj; c_tok[k] = bt;
      // d. stopping criteria: EOS or max_length reached
      c_hit[k] = (bt == st.eos) || (cur_len + 1 >= st.wmax[w]);
    }
    // e. running beams for the next iteration
    for (int k = 0; k < Kc; ++k) run_val[k] = c_val[k] + (c_hit[k] ? 1.0f : 0.0f) * -1.0e9f;
    unsigned used = 0;
    for (int i = 0; i < nb; ++i) {
      int bk = -1;
      for (int k = 0; k < Kc; ++k) if (!((used >> k) & 1u) && (bk < 0 || run_val[k] > run_val[bk])) bk = k;
      used |= 1u << bk;
      run_src[i] = c_beam[bk]; run_tok[i] = c_tok[bk]; n_run_score[i] = run_val[bk];
    }
    // f. finished beams
    const int un = st.unsat[w];
    const float denom = (float)pow((double)(cur_len + 1 - P), (double)st.length_penalty);
    for (int k = 0; k < Kc; ++k) {
      const int did = c_hit[k] && k < nb;
      float v = c_val[k] / denom;
      v = v + 0.0f * -1.0e9f;                       // beams_in_batch_are_full & early_stopping(False)
      v = v + (un ? 0.0f : 1.0f) * -1.0e9f;
      v = v + (did ? 0.0f : 1.0f) * -1.0e9f;
      fin_cand[k] = v;
    }
    unsigned usedm = 0;
    for (int i = 0; i < nb; ++i) {
      int bi = -1; float bv = 0.f;
      for (int m = 0; m < nb + Kc; ++m) {
        if ((usedm >> m) & 1u) continue;
        const float v = m < nb ? st.fin_score[w * nb + m] : fin_cand[m - nb];
        if (bi < 0 || v > bv) { bi = m; bv = v; }
      }
      usedm |= 1u << bi;
      fin_src[i] = bi;
      n_fin_score[i] = bv;
      n_fin_flag[i] = bi < nb ? st.fin_flag[w * nb + bi] : (c_hit[bi - nb] && (bi - nb) < nb);
      n_fin_len[i] = bi < nb ? st.fin_len[w * nb + bi] : cur_len + 1 - P;
    }
    // g. early-stop heuristic (early_stopping=False): can the best running beam still beat the worst finished one?
    const float best_running = n_run_score[0] / (float)pow((double)(cur_len + 1 - P), (double)st.length_penalty);
    float mn = n_fin_score[0];
    for (int i = 1; i < nb; ++i) mn = fminf(mn, n_fin_score[i]);
    int improve = 0;
    for (int i = 0; i < nb; ++i) { const float worst = n_fin_flag[i] ? mn : -1.0e9f; if (best_running > worst) improve = 1; }
    const int un_new = un && improve;
    st.unsat[w] = un_new;
    // the slot is finished once no running beam can still beat the finished ones (HF stops stepping a batch when this
    // holds for all of its items; until then the item's result is frozen) or max_length is reached
    if (!un_new || cur_len + 1 >= st.wmax[w]) st.done[w] = 1;
    else st.pos[w] = pos + 1;
    for (int i = 0; i < nb; ++i) {
      st.run_score[w * nb + i] = n_run_score[i];
      st.fin_score[w * nb + i] = n_fin_score[i];
      st.fin_flag[w * nb + i] = n_fin_flag[i];
      st.fin_len[w * nb + i] = n_fin_len[i];
      st.tokens_in[w * nb + i] = run_tok[i];
    }
  }
  __syncthreads();
  int* run = st.run_seq + (size_t)w * nb * L;
  int* fin = st.fin_seq + (size_t)w * nb * L;
  unsigned char* anc = st.anc + (size_t)w * nb * L;
  for (int p = lane; p < L; p += 64) {          // every position column is independent: read all, then write
    int rv[MAX_BEAMS], fv[MAX_BEAMS];
    unsigned char av[MAX_BEAMS];
#pragma unroll
    for (int i = 0; i < MAX_BEAMS; ++i) {
      if (i < nb) {
        const int src = run_src[i];
        rv[i] = p < cur_len ? run[src * L + p] : (p == cur_len ? run_tok[i] : st.pad);
        av[i] = p < cur_len ? anc[src * L + p] : (unsigned char)i;
        const int fs = fin_src[i];
        if (fs < nb) fv[i] = fin[fs * L + p];
        else { const int k = fs - nb; fv[i] = p < cur_len ? run[c_beam[k] * L + p] : (p == cur_len ? c_tok[k] : st.pad); }
      }
    }
#pragma unroll
    for (int i = 0; i < MAX_BEAMS; ++i) {
      if (i < nb) { run[i * L + p] = rv[i]; anc[i * L + p] = av[i]; fin[i * L + p] = fv[i]; }
    }
  }
}

__global__ void greedy_step_kernel(DecodeState st, const int* __restrict__ list, int n_list) {
  int w = blockIdx.x * 64 + threadIdx.x;
  if (list) { if (w >= n_list) return; w = list[w]; }
  if (w >= st.W || st.done[w]) return;
  const int L = st.L;
  const int pos = st.pos[w];
  const int cur_len = pos + 1;
  if (cur_len < st.P) {                 // prompt phase
    st.tokens_in[w] = st.prompt[cur_len];
    st.pos[w] = pos + 1;
    return;
  }
  int tok = st.cand_tok[(size_t)w * st.top_k];
  if (st.top_k > 1) {
    // HF TopKLogitsWarper + TopPLogitsWarper + multinomial over the top_k processed logits (sorted, best first):
    // candidate j survives the nucleus cut iff the probability mass of the candidates before it is < top_p
    const float* cv = st.cand_val + (size_t)w * st.top_k;
    const int* ct = st.cand_tok + (size_t)w * st.top_k;
    float pr[MAX_CAND];
    float z = 0.f;
    for (int j = 0; j < st.top_k; ++j) { pr[j] = expf(cv[j] - cv[0]); z += pr[j]; }      // -inf -> 0
    const bool cut = st.top_p > 0.f && st.top_p < 1.f;
    float kept = 0.f, before = 0.f;
    int n_keep = 0;
    for (int j = 0; j < st.top_k; ++j) {
      if (j > 0 && (pr[j] <= 0.f || (cut && before >= st.top_p * z))) break;
      kept += pr[j]; before += pr[j]; ++n_keep;
    }
    // counter-based uniform in [0, 1): splitmix64 of (seed, window, position)
    unsigned long long x = *st.seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)st.win[w] * 4096ull + (unsigned long long)cur_len + 1ull);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    const float u = (float)(x >> 40) * (1.0f / 16777216.0f) * kept;
    float acc = 0.f;
    tok = ct[n_keep - 1];
    for (int j = 0; j < n_keep; ++j) { acc += pr[j]; if (u < acc) { tok = ct[j]; break; } }
  }
  st.run_seq[(size_t)w * L + cur_len] = tok;
  st.anc[(size_t)w * L + cur_len] = 0;
  st.tokens_in[w] = tok;
  if (tok == st.eos || cur_len + 1 >= st.wmax[w]) {
    st.unsat[w] = 0;
    st.fin_len[w] = cur_len + 1 - st.P;
    st.done[w] = 1;
  } else {
    st.pos[w] = pos + 1;
  }
}

// One workgroup per retired slot.
__global__ void finalize_kernel(DecodeState st, const int* __restrict__ slots, int* out_tokens, int* out_lengths) {
  const int w = slots[blockIdx.x], L = st.L;
  const int row = st.win[w];
  if (row < 0) return;
  const int* src = (st.nb == 1 ? st.run_seq : st.fin_seq) + (size_t)w * st.nb * L;
  const int len = st.P + st.fin_len[w * st.nb];
  for (int p = threadIdx.x; p < L; p += blockDim.x) out_tokens[(size_t)row * L + p] = p < len ? src[p] : st.pad;
  if (threadIdx.x == 0) out_lengths[row] = len;
}

// ------------------------------------------------------------------------------------------------
int launch_decode_reset(const DecodeState& st, hipStream_t s) {
  hipLaunchKernelGGL(decode_reset_kernel, dim3(cdiv(st.W * st.nb, 256)), dim3(256), 0, s, st);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_decode_admit(const DecodeState& st, const int* slots, const int* wins, int n, int pf_np, int pos0, hipStream_t s) {
  if (n <= 0) return WSEG_OK;
  if (pf_np < 0 || pf_np > st.P || pos0 < 0 || pos0 >= st.P || pos0 > pf_np) { set_error("admission: %d prompt positions prefilled of %d, start %d", pf_np, st.P, pos0); return WSEG_ERR_INVALID; }
  hipLaunchKernelGGL(decode_admit_kernel, dim3(n), dim3(256), 0, s, st, slots, wins, pf_np, pos0);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_kv_assign(int* kv_pt, const int* pairs, int n, hipStream_t s) {
  if (n <= 0) return WSEG_OK;
  hipLaunchKernelGGL(kv_assign_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, kv_pt, pairs, n);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_decode_abort(const DecodeState& st, const int* slots, int n, hipStream_t s) {
  if (n <= 0) return WSEG_OK;
  hipLaunchKernelGGL(decode_abort_kernel, dim3(cdiv(n, 64)), dim3(64), 0, s, st, slots, n);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_build_suppress_mask(unsigned char* mask, int V, const int* sup, int n_sup, const int* bsup, int n_bsup, hipStream_t s) {
  hipLaunchKernelGGL(suppress_mask_kernel, dim3(1), dim3(1024), 0, s, mask, V, sup, n_sup, bsup, n_bsup);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_embed(int dtype, const DecodeState& st, const void* tok_emb, const void* pos_emb, void* x, int d, hipStream_t s) {
  const int R = st.W * st.nb;
#define WSEG_EMB(T_) hipLaunchKernelGGL((embed_kernel<T_>), dim3(R), dim3(256), 0, s, st, tok_emb, (const typename IO<T_>::P*)pos_emb, (float*)x, d)
  if (dtype == WSEG_BF16) WSEG_EMB(bf16_t);
  else if (dtype == WSEG_F16) WSEG_EMB(f16_t);
  else if (dtype == WSEG_BF16X3) WSEG_EMB(X3<bf16_t>);
  else if (dtype == WSEG_F16X3) WSEG_EMB(X3<f16_t>);
  else WSEG_EMB(float);
#undef WSEG_EMB
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_prompt_embed(int dtype, const DecodeState& st, int rows, int np, const void* tok_emb, const void* pos_emb, void* x, int d, hipStream_t s) {
#define WSEG_EMB(T_) hipLaunchKernelGGL((prompt_embed_kernel<T_>), dim3(rows), dim3(256), 0, s, st, np, tok_emb, (const typename IO<T_>::P*)pos_emb, (float*)x, d)
  if (dtype == WSEG_BF16X3) WSEG_EMB(X3<bf16_t>);
  else if (dtype == WSEG_F16X3) WSEG_EMB(X3<f16_t>);
  else if (dtype == WSEG_F32) WSEG_EMB(float);
  else { set_error("prompt pass: dtype %d", dtype); return WSEG_ERR_INVALID; }
#undef WSEG_EMB
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_prompt_self_attn(int dtype, const DecodeState& st, const float* qkv, const PartialInfo* qkv_part, const void* qkv_bias, void* kc, void* vc,
                            const int* slots, int n, int np, void* out, int H, int d, float scale, hipStream_t s) {
  if (np < 1 || np > 4 || np > KV_PAGE) { set_error("prompt pass: %d positions", np); return WSEG_ERR_INVALID; }
  if (!st.kv_pt) { set_error("prompt pass: page table missing"); return WSEG_ERR_STATE; }
  PartialInfo pi;
  if (qkv_part) pi = *qkv_part;
#define WSEG_PSA(TO_) hipLaunchKernelGGL((prompt_self_attn_kernel<TO_>), dim3(n * H), dim3(64), 0, s, qkv, pi, (const float*)qkv_bias, (float*)kc, (float*)vc, st.kv_pt, st.npg, slots, np, st.nb, H, d, out, scale)
  if (dtype == WSEG_BF16X3) WSEG_PSA(X3<bf16_t>);
  else if (dtype == WSEG_F16X3) WSEG_PSA(X3<f16_t>);
  else if (dtype == WSEG_F16M6) WSEG_PSA(M6);
  else { set_error("prompt pass: dtype %d", dtype); return WSEG_ERR_INVALID; }
#undef WSEG_PSA
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_dec_self_attn(int dtype, const DecodeState& st, const void* q, void* kc, void* vc, void* out, int H, int d,
                         const PartialInfo* qkv_part, const void* qkv_bias, float scale, hipStream_t s) {
  if (st.L > 512) { set_error("self-attention: max_length %d > 512", st.L); return WSEG_ERR_INVALID; }
  if (!st.kv_pt || st.npg * KV_PAGE < st.L) { set_error("self-attention: page table missing"); return WSEG_ERR_STATE; }
  const int R = st.W * st.nb;
  PartialInfo pi;
  if (qkv_part) pi = *qkv_part;
#define WSEG_SA(T_, TO_) hipLaunchKernelGGL((dec_self_attn_kernel<T_, TO_>), dim3(R * H), dim3(64), 0, s, st, (const T_*)q, (T_*)kc, (T_*)vc, out, H, d, pi, (const T_*)qkv_bias, scale)
  if (dtype == WSEG_BF16) WSEG_SA(bf16_t, bf16_t);
  else if (dtype == WSEG_F16) WSEG_SA(f16_t, f16_t);
  else if (dtype == WSEG_BF16X3) WSEG_SA(float, X3<bf16_t>);
  else if (dtype == WSEG_F16X3) WSEG_SA(float, X3<f16_t>);
  else if (dtype == WSEG_F16M6) WSEG_SA(float, M6);      // the o-proj GEMM's operand as M6 rows
  else WSEG_SA(float, float);
#undef WSEG_SA
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
template <typename T, typename TO>
static void launch_cross_t(const DecodeState& st, const void* q, const void* ck, const void* cv, void* out, int H, int Tk, int d,
                           const PartialInfo& pi, const void* qb, float scale, hipStream_t s) {
  dim3 grid(st.W * H), block(256);
#define WSEG_CA(NB_) hipLaunchKernelGGL((dec_cross_attn_kernel<T, TO, NB_>), grid, block, 0, s, st, (const T*)q, (const T*)ck, (const T*)cv, out, H, Tk, d, pi, (const T*)qb, scale)
  if (st.nb <= 1) WSEG_CA(1);
  else if (st.nb <= 2) WSEG_CA(2);
  else if (st.nb <= 4) WSEG_CA(4);
  else WSEG_CA(8);
#undef WSEG_CA
}
// wseg_kernels.h.  The three-MFMA modes' GEMMs are exact to ~6e-6 of a logit on the parity sweeps' models; the 24-bit FLOAT rows of
// r03 - mid r06 (format 1) added 2e-5 — which cost f16x3 one of 4 200 sweep recordings (a greedy decision with a margin of 2.8e-5;
// fp32 rows reproduce it) —, 16-bit block-floating-point rows 6e-5: since r06 these modes store 24-bit block floating point (format 3:
// the bytes of format 1 + 4 per row, error <= 2^-24 of the row maximum), with which f16x3 reproduces all 6 200 recordings of the seven
// sweeps.  The mixed mode's own fp6 cross terms cost 8e-5: it takes the 31 % smaller 16-bit rows (1.3e-4 in all).
// Knob builds: WSEG_X3_CKV = f32 | k24 | bfp | bfp24 for every split mode (attribution / A-B).
int x3_cross_kv_format(int dtype, int nb) {
  static const int forced = WSEG_KNOB_IS("WSEG_X3_CKV", "f32") ? 0 : (WSEG_KNOB_IS("WSEG_X3_CKV", "k24") ? 1 : (WSEG_KNOB_IS("WSEG_X3_CKV", "bfp") ? 2 : (WSEG_KNOB_IS("WSEG_X3_CKV", "bfp24") ? 3 : -1)));
  if (nb > 4) return 0;
  return forced >= 0 ? forced : (dtype == WSEG_F16M6 ? 2 : 3);
}
// WSEG_F16M6: does the cross-attention write its output (the co-proj GEMM's operand) as M6 rows?  The 24-bit K / V kernel does (it
// exists for up to 4 beams); 5..8 beams run the general fp32-K/V kernel, which writes hi | lo rows that the caller converts.
bool dec_cross_attn_writes_mx(int dtype, int nb) { return dtype == WSEG_F16M6 && x3_cross_kv_format(dtype, nb) != 0; }

int launch_dec_cross_attn(int dtype, const DecodeState& st, const void* q, const void* ck, const void* cv, void* out, int H, int Tk, int d,
                          const PartialInfo* q_part, const void* q_bias, float scale, hipStream_t s, const int* kv_slot) {
  if (Tk > 512) { set_error("cross-attention: %d encoder positions > 512", Tk); return WSEG_ERR_INVALID; }
  PartialInfo pi;
  if (q_part) pi = *q_part;
  const bool m6 = dtype == WSEG_F16M6;      // M6-row output from the 24-bit K / V kernel only (dec_cross_attn_writes_mx)
  const int kvf = (dtype == WSEG_BF16X3 || dtype == WSEG_F16X3 || m6) ? x3_cross_kv_format(dtype, st.nb) : 0;
  if (kvf == 2) {
    dim3 grid(st.W * H), block(256);
#define WSEG_BFP(TO_, NB_) hipLaunchKernelGGL((dec_cross_attn_bfp_kernel<TO_, NB_>), grid, block, 0, s, st, (const float*)q, (const unsigned char*)ck, (const unsigned char*)cv, out, H, Tk, d, pi, (const float*)q_bias, scale, kv_slot)
    if (dtype == WSEG_BF16X3) { if (st.nb <= 1) WSEG_BFP(X3<bf16_t>, 1); else if (st.nb <= 2) WSEG_BFP(X3<bf16_t>, 2); else WSEG_BFP(X3<bf16_t>, 4); }
    else if (m6) { if (st.nb <= 1) WSEG_BFP(M6, 1); else if (st.nb <= 2) WSEG_BFP(M6, 2); else WSEG_BFP(M6, 4); }
    else { if (st.nb <= 1) WSEG_BFP(X3<f16_t>, 1); else if (st.nb <= 2) WSEG_BFP(X3<f16_t>, 2); else WSEG_BFP(X3<f16_t>, 4); }
#undef WSEG_BFP
    WSEG_LAUNCH_CHECK();
    return WSEG_OK;
  }
  if (kvf == 3) {
    dim3 grid(st.W * H), block(256);
#define WSEG_K24(TO_, NB_) hipLaunchKernelGGL((dec_cross_attn_k24_kernel<TO_, NB_, true>), grid, block, 0, s, st, (const float*)q, (const unsigned char*)ck, (const unsigned char*)cv, out, H, Tk, d, pi, (const float*)q_bias, scale, kv_slot)
    if (dtype == WSEG_BF16X3) { if (st.nb <= 1) WSEG_K24(X3<bf16_t>, 1); else if (st.nb <= 2) WSEG_K24(X3<bf16_t>, 2); else WSEG_K24(X3<bf16_t>, 4); }
    else if (m6) { if (st.nb <= 1) WSEG_K24(M6, 1); else if (st.nb <= 2) WSEG_K24(M6, 2); else WSEG_K24(M6, 4); }
    else { if (st.nb <= 1) WSEG_K24(X3<f16_t>, 1); else if (st.nb <= 2) WSEG_K24(X3<f16_t>, 2); else WSEG_K24(X3<f16_t>, 4); }
#undef WSEG_K24
    WSEG_LAUNCH_CHECK();
    return WSEG_OK;
  }
  if (kvf == 1) {
    dim3 grid(st.W * H), block(256);
#define WSEG_K24(TO_, NB_) hipLaunchKernelGGL((dec_cross_attn_k24_kernel<TO_, NB_, false>), grid, block, 0, s, st, (const float*)q, (const unsigned char*)ck, (const unsigned char*)cv, out, H, Tk, d, pi, (const float*)q_bias, scale, kv_slot)
    if (dtype == WSEG_BF16X3) { if (st.nb <= 1) WSEG_K24(X3<bf16_t>, 1); else if (st.nb <= 2) WSEG_K24(X3<bf16_t>, 2); else WSEG_K24(X3<bf16_t>, 4); }
    else if (m6) { if (st.nb <= 1) WSEG_K24(M6, 1); else if (st.nb <= 2) WSEG_K24(M6, 2); else WSEG_K24(M6, 4); }
    else { if (st.nb <= 1) WSEG_K24(X3<f16_t>, 1); else if (st.nb <= 2) WSEG_K24(X3<f16_t>, 2); else WSEG_K24(X3<f16_t>, 4); }
#undef WSEG_K24
    WSEG_LAUNCH_CHECK();
    return WSEG_OK;
  }
  if (kv_slot) { set_error("cross-attention: the slot map exists for the 24-bit and block-floating-point K / V kernels only"); return WSEG_ERR_INVALID; }
  static const bool deep = getenv("WSEG_CROSS_NO_PK") == nullptr;         // tuning knob: fp32-FMA kernel
  if ((dtype == WSEG_BF16 || dtype == WSEG_F16) && deep && Tk <= 512 && st.nb <= 4) {
    dim3 grid(st.W * H), block(256);
#define WSEG_PK(HT_, NB_) hipLaunchKernelGGL((dec_cross_attn_pk_kernel<HT_, NB_>), grid, block, 0, s, st, (const HT_*)q, (const HT_*)ck, (const HT_*)cv, (HT_*)out, H, Tk, d, pi, (const HT_*)q_bias, scale)
    if (dtype == WSEG_BF16) { if (st.nb <= 1) WSEG_PK(bf16_t, 1); else if (st.nb <= 2) WSEG_PK(bf16_t, 2); else WSEG_PK(bf16_t, 4); }
    else { if (st.nb <= 1) WSEG_PK(f16_t, 1); else if (st.nb <= 2) WSEG_PK(f16_t, 2); else WSEG_PK(f16_t, 4); }
#undef WSEG_PK
  } else if (dtype == WSEG_BF16) launch_cross_t<bf16_t, bf16_t>(st, q, ck, cv, out, H, Tk, d, pi, q_bias, scale, s);
  else if (dtype == WSEG_F16) launch_cross_t<f16_t, f16_t>(st, q, ck, cv, out, H, Tk, d, pi, q_bias, scale, s);
  else if (dtype == WSEG_BF16X3) launch_cross_t<float, X3<bf16_t>>(st, q, ck, cv, out, H, Tk, d, pi, q_bias, scale, s);
  else if (dtype == WSEG_F16X3 || m6) launch_cross_t<float, X3<f16_t>>(st, q, ck, cv, out, H, Tk, d, pi, q_bias, scale, s);      // (f16m6, 5..8 beams: hi | lo rows)
  else launch_cross_t<float, float>(st, q, ck, cv, out, H, Tk, d, pi, q_bias, scale, s);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int row_topk_segments(int R) {
  int nseg = (512 + R - 1) / R;
  return nseg < 1 ? 1 : (nseg > 16 ? 16 : nseg);
}
template <int KC>
static void launch_topk_t(const DecodeState& st, const float* logits, float* pv, int* pi, float* ps, hipStream_t s, const int* list, int n_list) {
  // list mode (the admission's pass): the vocabulary is sliced as in a decode step of the whole slot population, so that the log-sum-exp
  // of a row is summed in the same order however many windows were admitted together
  const int R = (list ? n_list : st.W) * st.nb, nseg = row_topk_segments(st.W * st.nb);
  hipLaunchKernelGGL((row_topk_partial_kernel<KC>), dim3(nseg, R), dim3(256), 0, s, st, logits, nseg, pv, pi, ps, list);
  hipLaunchKernelGGL((row_topk_merge_kernel<KC>), dim3(R), dim3(64), 0, s, st, nseg, pv, pi, ps, list);
}
int launch_row_topk(const DecodeState& st, const float* logits, float* part_val, int* part_idx, float* part_stat, hipStream_t s,
                    const int* list, int n_list) {
  if (list && n_list <= 0) return WSEG_OK;
  const int Kc = st.nb == 1 ? st.top_k : 2 * st.nb;
  if (Kc == 1) launch_topk_t<1>(st, logits, part_val, part_idx, part_stat, s, list, n_list);
  else if (Kc <= 4) launch_topk_t<4>(st, logits, part_val, part_idx, part_stat, s, list, n_list);
  else if (Kc <= 8) launch_topk_t<8>(st, logits, part_val, part_idx, part_stat, s, list, n_list);
  else launch_topk_t<16>(st, logits, part_val, part_idx, part_stat, s, list, n_list);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_beam_step(const DecodeState& st, hipStream_t s, const int* list, int n_list) {
  if (list && n_list <= 0) return WSEG_OK;
  hipLaunchKernelGGL(beam_step_kernel, dim3(list ? n_list : st.W), dim3(64), 0, s, st, list);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_greedy_step(const DecodeState& st, hipStream_t s, const int* list, int n_list) {
  if (list && n_list <= 0) return WSEG_OK;
  hipLaunchKernelGGL(greedy_step_kernel, dim3(cdiv(list ? n_list : st.W, 64)), dim3(64), 0, s, st, list, n_list);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
int launch_finalize(const DecodeState& st, const int* slots, int n, int* out_tokens, int* out_lengths, hipStream_t s) {
  if (n <= 0) return WSEG_OK;
  hipLaunchKernelGGL(finalize_kernel, dim3(n), dim3(256), 0, s, st, slots, out_tokens, out_lengths);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

}  // namespace wseg

#ifdef WSEG_STAMPS
extern "C" int wseg_debug_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wseg::g_stamps), sizeof(unsigned long long) * 32);
}
#endif
