// GEMM kernels for gfx950:  C[m][n] = sum_k A[m][k] * W[n][k]  with fused epilogues.
//
// bf16 path (production): MFMA v_mfma_f32_16x16x32_bf16, LDS-staged through global_load_lds (16 B per
// lane, XOR-swizzled on the SOURCE address so the lane-linear LDS image is bank-conflict free for
// ds_read_b128), double-buffered with a counted vmcnt so the next K-tile streams in under the MFMAs.
// Operands are swapped (MFMA "A" = weight rows, "B" = activation rows) so that each lane ends up with
// 4 CONSECUTIVE output columns of one output row: bias / residual / stores are 8-byte vectors.
// Tile shapes: 128x128 (encoder, M = windows*500) and 64x64 / 32x64 with split-K (decoder steps, where
// M = windows*beams is small and the kernel is a weight stream bounded by HBM, not MFMA).
//
// f32 path (exact-parity mode): plain VALU 64x64 tile, fmaf chain in k order.
//
// Both paths share one epilogue (epi_apply), also used by the split-K reduction kernel.
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <type_traits>

#include <vector>
#include "wseg_gemm_epi.h"

namespace wseg {

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------
#define WSEG_GLDS16(gptr, ldsptr)                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                   (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)
// the same with the non-temporal policy (aux = 2): for weight slices that exactly ONE workgroup reads once — the decode steps of a few
// windows (MI355X_MICROARCH.md, row nt-weights: issued -> landed -18 %, a decode layer -5..10 %); never where several workgroups
// re-read the slice from L2 (there nt measured -6 % end to end)
#define WSEG_GLDS16_NT(gptr, ldsptr)                                                           \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                   (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 2)

// Measurement builds only (python -m whisperseg_amd.build --stamps 4, tools/pp_stamps.py): the ping-pong kernel's workgroup 0 records
// the shader clock (s_memtime) of wave 0 (row group 0) and wave 4 (row group 1) around the L and M parts of both phases of K tiles 8..11.
#if defined(WSEG_STAMPS) && WSEG_STAMPS == 4
__device__ unsigned long long g_pp_stamps[2 * 4 * 4 * 4 + 4];  // [group][K tile 8..11][phase][top, L issued, M start, M end]; then (cycles, 100-MHz ticks) at K tiles 2 and 18
#define WSEG_PP_STAMP(P, I) do { if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && g >= 8 && g < 12) \
    g_pp_stamps[((wr * 4 + (g - 8)) * 4 + (P)) * 4 + (I)] = __builtin_readcyclecounter(); } while (0)
// effective shader clock under this kernel's own load: the cycle counter against the constant 100-MHz counter over 16 K tiles
#define WSEG_PP_CLOCK() do { if (blockIdx.x == 0 && wave == 0 && lane == 0 && (g == 2 || g == 18)) { \
    g_pp_stamps[128 + (g == 18 ? 2 : 0)] = __builtin_readcyclecounter(); g_pp_stamps[129 + (g == 18 ? 2 : 0)] = wall_clock64(); } } while (0)
#else
#define WSEG_PP_STAMP(P, I) do { } while (0)
#define WSEG_PP_CLOCK() do { } while (0)
#endif

#ifndef WSEG_VT_TRANSPOSED
#define WSEG_VT_TRANSPOSED 1      // (0: the row-wise 2-byte V^T stores of r01-r05, for same-box A/B builds)
#endif
#ifndef WSEG_SKINNY_NT
#define WSEG_SKINNY_NT 1      // (0: same-box A/B builds)
#endif
#ifndef WSEG_PP_LATEWAIT
#define WSEG_PP_LATEWAIT 1
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Barrier that closes the fragment reads of an LDS stage: every ds_read this wave has issued must have RETURNED before
// the wave signals that the stage may be refilled.  A bare s_barrier only orders instruction issue; the compiler is
// free to (and did) leave the last fragment reads in flight across it, and an LDS-DMA write of the next K tile issued by
// another wave right after the barrier can then land before those reads execute.  On an otherwise idle chip the reads
// win by a wide margin; with load-heavy work sharing the CU (a second stream, or this kernel's own fp32-residual
// epilogue in a co-resident workgroup) they lost about once per 10^4 K tiles (tools/concurrency_stress.py).
__device__ __forceinline__ void lds_reads_done_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MX fragment reads (M6 rows, wseg_common.h).  A lane (fr = lane & 15, fg = lane >> 4) reads, from row fr of a 16-row block of a
// [rows][64 words] operand tile (128-byte rows, 16-byte slots XOR-swizzled with row & 7 like every operand tile), its 32-byte
// chunk fg = logical slots 2 fg and 2 fg + 1: two ds_read_b128.  The two byte offsets depend on the lane only (MxOff, computed
// once per kernel); a 16-row block is 2048 bytes further on.
struct MxOff { int p0, p1; };
__device__ __forceinline__ MxOff mx_offsets(int fr, int fg) {
  const int sw = fr & 7, base = fr * 128;
  MxOff m;
  m.p0 = base + (((2 * fg) ^ sw) << 4);
  m.p1 = base + (((2 * fg + 1) ^ sw) << 4);
  return m;
}
__device__ __forceinline__ MxFrag ld_mx_frag(const void* block16, const MxOff& m) {
  const unsigned char* b = (const unsigned char*)block16;
  const uint4 p0 = *(const uint4*)(b + m.p0), p1 = *(const uint4*)(b + m.p1);
  MxFrag f;
  f.v = (mx_i32x8){(int)p0.x, (int)p0.y, (int)p0.z, (int)p0.w, (int)p1.x, (int)p1.y, (int)p1.z, (int)p1.w};
  return f;
}
// The same for the ping-pong kernel, which runs at the 256-register cap: the second read as ds_read_b96 (codes 4, 5 + scale: the
// chunk's padding dword never takes a register), 7 registers per fragment.  Opaque to the compiler's waitcnt bookkeeping: the
// caller retires the reads with its own s_waitcnt lgkmcnt(0) (the L part of a phase ends with one anyway).
// The ping-pong kernel runs at the 256-register cap, and the MFMA builtin wants 8-register operands (two of them dead for fp6):
// there the fragment is exactly what the instruction reads — a 6-register tuple of codes + one scale register (b128 + b96 LDS
// reads: the chunk's padding dword never takes a register) — and the instruction is issued through inline assembly.  Its
// accumulator is written by builtin MFMAs and asm MFMAs alternately, always with a barrier and a round of LDS reads in between.
typedef unsigned mx_u2 __attribute__((ext_vector_type(2)));
typedef unsigned mx_u3 __attribute__((ext_vector_type(3)));
typedef unsigned mx_u4 __attribute__((ext_vector_type(4)));
struct MxFrag7 { mx_u6 v; unsigned s; };
typedef const char __attribute__((address_space(3))) * lds_cp;
// p0 / p1: LDS byte pointers of the lane's two 16-byte slots in 16-row block 0 (operand tile base + MxOff); OFF (a multiple of the
// 2048-byte block) lands in the instruction's offset field — pointer arithmetic, no address VALU per read
template <int OFF>
__device__ __forceinline__ MxFrag7 ld_mx_frag7(lds_cp p0, lds_cp p1) {
  typedef const mx_u4 __attribute__((address_space(3))) * lds_u4;
  typedef const mx_u2 __attribute__((address_space(3))) * lds_u2;
  typedef const unsigned __attribute__((address_space(3))) * lds_u1;
  // b128 + b64 + b32: every load fills a whole sub-range of the operand tuple (codes 0..3 | 4..5) or the scale register — no register
  // copies (a b96 for codes 4, 5 + scale needed two v_mov per fragment) and 8 instead of 12 LDS cycles
  const mx_u4 q0 = *(lds_u4)(p0 + OFF);
  const mx_u2 q1 = *(lds_u2)(p1 + OFF);
  MxFrag7 f;
  f.v = (mx_u6){q0[0], q0[1], q0[2], q0[3], q1[0], q1[1]};
  f.s = *(lds_u1)(p1 + OFF + 8);
  return f;
}
__device__ __forceinline__ void mfma_mx6_asm(f32x4& c, const MxFrag7& a, const MxFrag7& b) {
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:2 blgp:2"
               : "+v"(c) : "v"(a.v), "v"(b.v), "v"(a.s), "v"(b.s));
}

// SPLIT: write fp32 partials [z][m_pad][n] (epilogue applied later by splitk_reduce_kernel).
// T = X3<HT> (split-precision modes): A / W rows are hi | lo pairs (wseg_common.h), lda / ldw / k_len count 16-bit words
// (twice the logical K), and a K tile is multiplied as (W hi, A hi) + (W hi, A lo) + (W lo, A hi).
template <typename T, int BM, int BN, int WM, int WN, int EPI, bool SPLIT, int NST = 2>
__global__ __launch_bounds__(WM * WN * 64, (BM <= 128 && BN <= 64 && WM * WN == 4) ? 3 : 1) void gemm_h16_kernel(const typename IO<T>::H* __restrict__ A, int lda,
                                                        const typename IO<T>::H* __restrict__ W, int ldw,
                                                        int M, int N, int k_len, EpiParams ep,
                                                        float* __restrict__ part, int m_pad, int ntm) {
  typedef typename IO<T>::H HT;
  constexpr bool MXM = IsMx<T>::v;               // M6 rows: K tiles alternate hi (plain half MFMAs) / MX (one scaled MFMA per 16x16 tile)
  constexpr bool X3M = IO<T>::split && !MXM;
  constexpr int BK = 64;
  constexpr int TM = BM / WM, TN = BN / WN;      // wave tile
  constexpr int MI = TM / 16, NI = TN / 16;      // 16x16 MFMA tiles per wave
  constexpr int NT = WM * WN * 64;               // threads per workgroup (4 or 8 waves)
  constexpr int A_IT = BM * 8 / NT, W_IT = BN * 8 / NT;
  constexpr int NLD = A_IT + W_IT;
  static_assert((WM * WN == 4 || WM * WN == 8) && A_IT >= 1 && W_IT >= 1, "tile config");
  __shared__ __attribute__((aligned(16))) HT smem[NST * (BM + BN) * BK];
  HT* sA = smem;                     // [NST][BM*64]
  HT* sW = smem + NST * BM * BK;     // [NST][BN*64]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  if (ntm > 0) {
    // 1-D grid, XCD-aware: workgroup b runs on XCD b % 8 (observed dispatch order, speed only), so give
    // every XCD a contiguous run of tiles (bijective remap), ordered m-fastest inside groups of 8 m-tiles:
    // neighbours in time and on the same L2 share the weight tile and 8 activation tiles.
    const int nblk = gridDim.x, bid = blockIdx.x, ntn = N / BN;
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, loc = bid >> 3;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    constexpr int GM = 8;
    const int per_group = GM * ntn, grp = swz / per_group, rem = swz - grp * per_group;
    const int gm = min(GM, ntm - grp * GM);
    m0 = (grp * GM + rem % gm) * BM;
    n0 = (rem / gm) * BN;
  }
  const int kbeg = blockIdx.z * k_len;
  const int nk = k_len / BK;

  // per-thread source pointers (swizzle on the source: LDS slot p holds logical 16-B slot (p&7)^(row&7))
  const HT* a_src[A_IT];
  const HT* w_src[W_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int p = it * NT + tid, row = p >> 3, sl = (p & 7) ^ (row & 7);
    a_src[it] = A + (size_t)(m0 + row) * lda + kbeg + sl * 8;
  }
#pragma unroll
  for (int it = 0; it < W_IT; ++it) {
    const int p = it * NT + tid, row = p >> 3, sl = (p & 7) ^ (row & 7);
    w_src[it] = W + (size_t)(n0 + row) * ldw + kbeg + sl * 8;
  }
  // one row tile (2-D grid): every (column tile, K range) of W is read by exactly this workgroup, once -> non-temporal weight stream
  // (r06, same-box A/B at 8 / 15 / 32 windows x 4 beams, f16m6: 3.03 -> 2.94, 3.47 -> 3.45, 4.40 -> 4.46 ms per decode step: the 128-row
  // tile keeps the default policy)
  const bool w_once = WSEG_SKINNY_NT && BN == 64 && BM <= 64 && ntm == 0 && gridDim.y == 1;
  auto issue = [&](int kt, int buf) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it)
      WSEG_GLDS16(a_src[it] + kt * BK, sA + buf * BM * BK + (it * NT + wave * 64) * 8);
    if (w_once) {
#pragma unroll
      for (int it = 0; it < W_IT; ++it)
        WSEG_GLDS16_NT(w_src[it] + kt * BK, sW + buf * BN * BK + (it * NT + wave * 64) * 8);
    } else {
#pragma unroll
      for (int it = 0; it < W_IT; ++it)
        WSEG_GLDS16(w_src[it] + kt * BK, sW + buf * BN * BK + (it * NT + wave * 64) * 8);
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  // one K tile (64) of the wave tile from LDS stage `buf`.  Fragment reads are software-pipelined against the MFMAs: the
  // activation fragments of BOTH k-steps of the tile and the next weight fragment are requested while the current weight
  // fragment is being multiplied, so a wave only stalls on LDS at the head of a tile.
  auto compute = [&](int buf) {
    const HT* cA = sA + buf * BM * BK + (wm * TM) * BK;
    const HT* cW = sW + buf * BN * BK + (wn * TN) * BK;
    auto lda = [&](int kk, int j) {
      const int r = j * 16 + fr;
      return *(const bf16x8*)(cA + r * BK + (((kk * 4 + fg) ^ (r & 7)) << 3));
    };
    auto ldw = [&](int kk, int i) {
      const int r = i * 16 + fr;
      return *(const bf16x8*)(cW + r * BK + (((kk * 4 + fg) ^ (r & 7)) << 3));
    };
    bf16x8 af0[MI], af1[MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) af0[j] = lda(0, j);
    if constexpr (X3M) {                       // the lo halves of the activations are multiplied with the hi weights
#pragma unroll
      for (int j = 0; j < MI; ++j) af1[j] = lda(1, j);
    }
    bf16x8 wcur = ldw(0, 0);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bf16x8 wnxt = (i + 1 < NI) ? ldw(0, i + 1) : ldw(1, 0);
      if (!X3M && i == NI - 1) {
#pragma unroll
        for (int j = 0; j < MI; ++j) af1[j] = lda(1, j);
      }
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = H16<HT>::mfma16(wcur, af0[j], acc[i][j]);
      if constexpr (X3M) {
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = H16<HT>::mfma16(wcur, af1[j], acc[i][j]);
      }
      wcur = wnxt;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      bf16x8 wnxt = wcur;
      if (i + 1 < NI) wnxt = ldw(1, i + 1);
      // every fragment of this K tile has been requested: retire the reads and release the stage, then multiply the last
      // group (which needs the retired reads anyway) while the other waves arrive
      if (i == NI - 1) lds_reads_done_barrier();
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = H16<HT>::mfma16(wcur, X3M ? af0[j] : af1[j], acc[i][j]);
      wcur = wnxt;
    }
  };
  // M6 rows, MX tile: both cross terms of 64 logical columns in one scaled MFMA per 16x16 tile
  auto compute_mx = [&](int buf) {
    const HT* cA = sA + buf * BM * BK + (wm * TM) * BK;
    const HT* cW = sW + buf * BN * BK + (wn * TN) * BK;
    const MxOff mo = mx_offsets(fr, fg);
    MxFrag am[MI];
#pragma unroll
    for (int j = 0; j < MI; ++j) am[j] = ld_mx_frag(cA + j * 16 * BK, mo);
    MxFrag wcur = ld_mx_frag(cW, mo);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      MxFrag wnxt = wcur;
      if (i + 1 < NI) wnxt = ld_mx_frag(cW + (i + 1) * 16 * BK, mo);
      if (i == NI - 1) lds_reads_done_barrier();      // every fragment of the tile has been read: the stage may be refilled
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = mfma_mx6(wcur, am[j], acc[i][j]);
      wcur = wnxt;
    }
  };
  // NST-deep LDS ring.  NST == 2: prefetch one tile ahead.  NST > 2 (decoder weight streams, few K tiles per
  // workgroup): NST-1 tiles are kept in flight; past the end the last tile is re-issued so the counted vmcnt
  // stays a compile-time constant.
  if constexpr (NST == 2) {
    issue(0, 0);
  } else {
#pragma unroll
    for (int t = 0; t < NST - 1; ++t) issue(t < nk ? t : nk - 1, t);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt % NST;
    if constexpr (NST == 2) {
      if (kt + 1 < nk) {
        issue(kt + 1, buf ^ 1);
        wait_vmcnt<NLD>();
      } else {
        wait_vmcnt<0>();
      }
    } else {
      const int nt = kt + NST - 1;
      issue(nt < nk ? nt : nk - 1, nt % NST);
      wait_vmcnt<NLD*(NST - 1)>();
    }
    __builtin_amdgcn_s_barrier();
    if (MXM && ((kbeg / BK + kt) & 1)) compute_mx(buf);
    else compute(buf);                              // ends with lds_reads_done_barrier(): the stage may be refilled
  }

  // epilogue: lane holds n = nb + i*16 + fg*4 + {0..3}, m = mb + j*16 + fr
  if constexpr (!SPLIT && TN == 64) {
    // Staged through LDS (free after the last barrier of the K loop): every wave transposes one 16-row x 64-column
    // strip at a time so that 8 lanes cover one 128-byte output row with 16-byte accesses (bias, residual, store) —
    // instead of 8-byte accesses scattered over 16 rows per instruction.
    constexpr int LDT = TN + 4;                       // padded row stride (floats): conflict-free b128 writes
    float* strip = (float*)smem + (size_t)wave * 16 * LDT;
    const int rr = lane >> 3, cc = (lane & 7) * 8;
#pragma unroll
    for (int j = 0; j < MI; ++j) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
        *(float4*)(strip + fr * LDT + i * 16 + fg * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      // same-wave LDS RAW: ds ops of one wave complete in order, the compiler waits lgkmcnt before the reads
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int r = hh * 8 + rr;
        const float4 a = *(const float4*)(strip + r * LDT + cc), b = *(const float4*)(strip + r * LDT + cc + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const int m = m0 + wm * TM + j * 16 + r;
        if (m < M) epi_apply8<EPI, T>(ep, m, n0 + wn * TN + cc, v);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int j = 0; j < MI; ++j) {
        const int m = m0 + wm * TM + j * 16 + fr;
        const int n = n0 + wn * TN + i * 16 + fg * 4;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if constexpr (SPLIT) {
          *(float4*)(part + ((size_t)blockIdx.z * m_pad + m) * N + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          if (m < M) epi_apply<EPI, T>(ep, m, n, v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Epilogue of the 8-wave MFMA kernels (wave tile MI*16 rows x 64 columns), staged through LDS: every wave
// transposes one 16-row x 64-column strip at a time so that 8 lanes cover one 128-byte output row with 16-byte
// accesses.  Every global LOAD (bias, residual rows) is issued before the first STORE: vmcnt retires in order on
// gfx9, so a load issued after a store cannot be consumed before that store has completed — a bias load per
// output row made each of the 16 stores a full round trip (measured 16-42 % of a launch).
// ------------------------------------------------------------------------------------------------
// (NIT / I0: the accumulator array holds NIT column tiles of which I0 .. I0 + 3 are written — the 128x128 wave tiles of gemm_w4_kernel
// go out as two 64-column halves)
#ifndef WSEG_EPI_M6_LDS
#define WSEG_EPI_M6_LDS 1      // (0: the register-only row writer + scalar GELU of r04, for same-box A/B builds)
#endif
template <typename T, int EPI, int MI, int NI, int NIT = NI, int I0 = 0>
__device__ __forceinline__ void staged_epilogue(const f32x4 (&acc)[NIT][MI], float* stage, const EpiParams& ep, int M, int mb,
                                                int nb, int lane, int wave) {
  static_assert(NI == 4, "64-column wave tiles");
  typedef typename IO<T>::P PT;
  constexpr int LDT = 64 + 4;
  float* strip = stage + (size_t)wave * 16 * LDT;
  const int fr = lane & 15, fg = lane >> 4, rr = lane >> 3, cc = (lane & 7) * 8;
  const int nc = nb + cc;
  float bv[8];
  if (ep.bias) ld8_h<PT>((const PT*)ep.bias + nc, bv);
  else {
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
  }
  // fp32 residual rows: NS rows (8 floats = two 16-byte loads each) are in flight per lane.  All of them are requested
  // before the first store; row t + NS is requested right after row t has been stored, NS - 1 stores ahead of its use, so
  // it never waits for a store it was issued behind (vmcnt retires in order).
  constexpr int NROW = MI * 2, NS = NROW < 8 ? NROW : 8;
  float4 rres[EPI == EPI_RESID ? NS : 1][2];
  const float* rbase = (const float*)ep.resid + nc;
  if constexpr (EPI == EPI_RESID) {
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      const float* rp = rbase + (size_t)min(mb + t * 8 + rr, M - 1) * ep.ldc;
      rres[t][0] = *(const float4*)rp; rres[t][1] = *(const float4*)(rp + 4);
    }
  }
  EpiParams ep2 = ep;
  ep2.bias = nullptr;
  // cross-K/V into window slots: the (at most two: 16 * MI <= t_len rows) windows this wave tile touches are looked up here,
  // before the first store (see above), and rows are re-based so that the identity mapping lands in the slot
  int kv_b0 = 0, kv_s0 = 0, kv_s1 = 0;
  if constexpr (EPI == EPI_KV_CROSS) {
    if (ep.slot_map) {
      const int nwin = (M + ep.t_len - 1) / ep.t_len;
      kv_b0 = min(mb, M - 1) / ep.t_len;
      kv_s0 = ep.slot_map[kv_b0];
      kv_s1 = ep.slot_map[min(kv_b0 + 1, nwin - 1)];
      ep2.slot_map = nullptr;
    }
  }
  // Decoder q | k | v into the PAGED self-attention cache (EPI_QKV_DEC): a row's destination needs three dependent lookups (idle flag,
  // position, page-table entry).  Done row by row inside the store loop (r04-r05) every lookup queued behind the previous row's stores
  // (vmcnt retires in order: a load issued after a store waits for it) — 16 serialised round trips per wave tile, ~30 of the 87 us of the
  // 4 096-row launch.  Now all 16 rows of the lane are resolved before the first store (see the rule above).
  [[maybe_unused]] int dec_kv[EPI == EPI_QKV_DEC ? NROW : 1];      // ((unit * beams + beam) * heads) * KV_PAGE + pos % KV_PAGE, or -1: no store
  [[maybe_unused]] int dec_sec = 0;
  if constexpr (EPI == EPI_QKV_DEC) {
    dec_sec = nb / ep.d_model;      // wave-uniform: a 64-column wave tile lies inside one section and one head
    if (dec_sec > 0) {
      int pos_r[NROW], slot_r[NROW];
#pragma unroll
      for (int t = 0; t < NROW; ++t) {
        const int m = mb + t * 8 + rr;
        slot_r[t] = min(m, M - 1) / ep.pos_div;
        dec_kv[t] = (m < M && !ep.idle_ptr[slot_r[t]]) ? 0 : -1;
        pos_r[t] = ep.pos_ptr[slot_r[t]];
      }
#pragma unroll
      for (int t = 0; t < NROW; ++t) {
        const int m = min(mb + t * 8 + rr, M - 1), beam = m - slot_r[t] * ep.pos_div;
        const int unit = ep.kv_pt[(size_t)slot_r[t] * ep.kv_npg + pos_r[t] / KV_PAGE];
        if (dec_kv[t] == 0) dec_kv[t] = ((unit * ep.pos_div + beam) * ep.n_heads) * KV_PAGE + (pos_r[t] % KV_PAGE);
      }
    }
  }
  // V section of the fused encoder q | k | v GEMM (EPI_QKV_ENC, 16-bit V^T planes [b][h][64][t_pad]): the strip is read TRANSPOSED, a lane
  // takes 4 consecutive positions t of one head dimension — one 8-byte store per plane where the row-wise path issues four 2-byte
  // stores (r06: the 2-byte stores were 256 store instructions per wave tile, 8 x 16-byte segments each; now 64 instructions whose
  // 16 lanes-of-4 cover 32 contiguous bytes per V^T row).  Window starts are multiples of 4 rows (t_len % 4 == 0, checked), so a
  // granule never straddles two windows.  vt_fast is wave-uniform.
  [[maybe_unused]] bool vt_fast = false;
  [[maybe_unused]] float vt_bias[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == EPI_QKV_ENC && WSEG_VT_TRANSPOSED) {
    typedef typename IO<T>::A AT;
    vt_fast = sizeof(AT) == 2 && nb >= 2 * ep.d_model && (ep.t_len & 3) == 0 && (M & 3) == 0 && !(IO<T>::split && ep.qkv_mode == 1);
    if (vt_fast && ep.bias) {
#pragma unroll
      for (int q = 0; q < 4; ++q) vt_bias[q] = El<PT>::ld((const PT*)ep.bias + nb + q * 16 + (lane >> 2));
    }
  }
#pragma unroll
  for (int j = 0; j < MI; ++j) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      *(float4*)(strip + fr * LDT + i * 16 + fg * 4) = make_float4(acc[I0 + i][j][0], acc[I0 + i][j][1], acc[I0 + i][j][2], acc[I0 + i][j][3]);
    // same-wave LDS RAW: ds ops of one wave complete in order, the compiler waits lgkmcnt before the reads
    if constexpr (EPI == EPI_QKV_ENC && WSEG_VT_TRANSPOSED) {
      if (vt_fast) {
        typedef typename IO<T>::A AT;
        const int g4 = lane & 3, hq = lane >> 2;
        const int m = mb + j * 16 + g4 * 4;                      // first of the lane's 4 rows
        const int b = m / ep.t_len, t = m - b * ep.t_len;
        const int nn = nb - 2 * ep.d_model, h = nn >> 6;
        AT* vbase = (AT*)ep.v + ((size_t)b * ep.n_heads + h) * 64 * ep.t_pad;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int hd = q * 16 + hq;
          float x[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = strip[(g4 * 4 + r) * LDT + hd] + vt_bias[q];
          if constexpr (IO<T>::split) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = H16<AT>::sat(x[r]);
          }
          const uint2 hi2 = make_uint2(H16<AT>::pack(x[0], x[1]), H16<AT>::pack(x[2], x[3]));
          AT* dst = vbase + (ep.vt_tiled ? vt_tiled_index(hd, t) : hd * ep.t_pad + t);      // 4 consecutive keys are contiguous either way
          if (m < M) *(uint2*)dst = hi2;
          if (IO<T>::split && ep.qkv_mode == 2) {               // lo plane: x - rn(x)
            const uint2 lo2 = make_uint2(H16<AT>::pack(H16<AT>::sub_lo(x[0], hi2.x), H16<AT>::sub_hi(x[1], hi2.x)),
                                         H16<AT>::pack(H16<AT>::sub_lo(x[2], hi2.y), H16<AT>::sub_hi(x[3], hi2.y)));
            if (m < M) *(uint2*)(dst + ep.qkv_plane) = lo2;
          }
        }
        continue;
      }
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int rw = hh * 8 + rr;
      const float4 a = *(const float4*)(strip + rw * LDT + cc), b = *(const float4*)(strip + rw * LDT + cc + 4);
      float v[8] = {a.x + bv[0], a.y + bv[1], a.z + bv[2], a.w + bv[3], b.x + bv[4], b.y + bv[5], b.z + bv[6], b.w + bv[7]};
      const int m = mb + j * 16 + rw;
      if constexpr (EPI == EPI_RESID) {
        const int t = j * 2 + hh, sl = t % NS;
        const float4 r0 = rres[sl][0], r1 = rres[sl][1];
        if (m < M) {
          float* o = (float*)ep.out + (size_t)m * ep.ldc + nc;
          *(float4*)o = make_float4(r0.x + v[0], r0.y + v[1], r0.z + v[2], r0.w + v[3]);
          *(float4*)(o + 4) = make_float4(r1.x + v[4], r1.y + v[5], r1.z + v[6], r1.w + v[7]);
        }
        if (t + NS < NROW) {
          const float* rp = rbase + (size_t)min(mb + (t + NS) * 8 + rr, M - 1) * ep.ldc;
          rres[sl][0] = *(const float4*)rp; rres[sl][1] = *(const float4*)(rp + 4);
        }
      } else if constexpr (EPI == EPI_KV_CROSS) {
        int mm = m;
        if (ep.slot_map) { const int b = m / ep.t_len; mm = (b == kv_b0 ? kv_s0 : kv_s1) * ep.t_len + (m - b * ep.t_len); }
        if (m < M) epi_apply8<EPI, T>(ep2, mm, nc, v);
      } else if constexpr (EPI == EPI_QKV_DEC) {
        const int nn = nc - dec_sec * ep.d_model;
        if (dec_sec == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= ep.scale;
          if (m < M) st8_h<PT>((PT*)ep.q + (size_t)m * ep.d_model + nn, v);
        } else {
          const int kv = dec_kv[j * 2 + hh];
          if (kv >= 0) st8_h<PT>((PT*)(dec_sec == 1 ? ep.k : ep.v) + ((size_t)kv + (size_t)(nn >> 6) * KV_PAGE) * 64 + (nn & 63), v);
        }
      } else if constexpr (WSEG_EPI_M6_LDS && IsMx<T>::v && (EPI == EPI_GELU || EPI == EPI_STORE)) {
        // M6 rows: the quad's word exchange goes through the strip slots the quad has just read (8 floats per lane = its 128 bytes)
        if constexpr (EPI == EPI_GELU) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) { const gelu_f2 y = gelu_erf_fast2((gelu_f2){v[e], v[e + 1]}); v[e] = y[0]; v[e + 1] = y[1]; }
        }
        op_st8_m6_lds(ep.out, (size_t)m, ep.ldc, nc, v, (unsigned char*)(strip + rw * LDT + (cc & ~31)), m < M);
      } else {
        if (m < M) epi_apply8<EPI, T>(ep2, m, nc, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Persistent variant for the large encoder GEMMs: the grid is one (256x256) or two (128x128) workgroups per CU and
// every workgroup walks a strided sequence of tiles.  While the last K-tile of a tile is multiplied the first
// K-tile of the NEXT tile is already streaming into the other LDS stage, and it keeps streaming under the LDS-staged
// epilogue — with 20 K-tiles per tile (K = 1280) the exposed fill + epilogue were ~25 % of a tile.
// Tile order: XCD x (workgroup id % 8, observed dispatch, speed only) owns a contiguous run of the m-fastest /
// 8-row-group tile order; its workgroups interleave over that run, so tiles in flight on one L2 are neighbours.
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM * WN * 64, (BM == 128 && BN == 128) ? 2 : 1) void gemm_h16_persist_kernel(const typename IO<T>::H* __restrict__ A, int lda,
                                                                         const typename IO<T>::H* __restrict__ W, int ldw, int M, int N,
                                                                         int K, EpiParams ep, int ntm, int GM) {
  typedef typename IO<T>::H HT;
  constexpr bool MXM = IsMx<T>::v;
  constexpr bool X3M = IO<T>::split && !MXM;
  constexpr int BK = 64;
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int NT = WM * WN * 64;
  constexpr int A_IT = BM * 8 / NT, W_IT = BN * 8 / NT;
  constexpr int NLD = A_IT + W_IT;
  constexpr int STAGE = (BM + BN) * BK;            // elements per LDS stage: [A tile | W tile]
  static_assert(TN == 64, "staged epilogue needs 64-column wave tiles");
  __shared__ __attribute__((aligned(16))) HT smem[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 15, fg = lane >> 4;
  const int ntn = N / BN, ntiles = ntm * ntn, nk = K / BK;
  // this workgroup's tile sequence
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int count = q + (xcd < r ? 1 : 0);
  auto tile_coords = [&](int swz, int& m0, int& n0) {
    const int per_group = GM * ntn, grp = swz / per_group, rem = swz - grp * per_group;
    const int gm = min(GM, ntm - grp * GM);
    m0 = (grp * GM + rem % gm) * BM;
    n0 = (rem / gm) * BN;
  };
  const HT* a_src[A_IT];
  const HT* w_src[W_IT];
  auto set_src = [&](int m0, int n0) {
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int p = it * NT + tid, row = p >> 3, sl = (p & 7) ^ (row & 7);
      a_src[it] = A + (size_t)(m0 + row) * lda + sl * 8;
    }
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
      const int p = it * NT + tid, row = p >> 3, sl = (p & 7) ^ (row & 7);
      w_src[it] = W + (size_t)(n0 + row) * ldw + sl * 8;
    }
  };
  auto issue = [&](int kt, int buf) {
    HT* st = smem + buf * STAGE;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) WSEG_GLDS16(a_src[it] + kt * BK, st + (it * NT + wave * 64) * 8);
#pragma unroll
    for (int it = 0; it < W_IT; ++it) WSEG_GLDS16(w_src[it] + kt * BK, st + BM * BK + (it * NT + wave * 64) * 8);
  };

  int idx = loc;
  if (idx >= count) return;
  int m0, n0;
  tile_coords(start + idx, m0, n0);
  set_src(m0, n0);
  issue(0, 0);
  int g = 0;                                        // K-tiles consumed so far: stage parity
  while (true) {
    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nidx = idx + bpx;
    const bool has_next = nidx < count;
    int nm0 = 0, nn0 = 0;
    if (has_next) tile_coords(start + nidx, nm0, nn0);
    // tile bodies as lambdas: the M6 instantiation runs a (hi tile, MX tile) pair per iteration as straight-line code (register
    // allocation: see gemm_h16_pp_kernel)
    auto tile_head = [&](int kt, const HT*& cA, const HT*& cW) {
      const int buf = (g + kt) & 1;
      if (kt + 1 < nk) {
        issue(kt + 1, buf ^ 1);
        wait_vmcnt<NLD>();
      } else if (has_next) {
        set_src(nm0, nn0);                          // all loads of the current tile are issued: retarget
        issue(0, buf ^ 1);
        wait_vmcnt<NLD>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      cA = smem + buf * STAGE + (wm * TM) * BK;
      cW = smem + buf * STAGE + BM * BK + (wn * TN) * BK;
    };
    auto hi_tile = [&](int kt) {
      const HT *cA, *cW;
      tile_head(kt, cA, cW);
      auto lda_f = [&](int kk, int j) {
        const int rw = j * 16 + fr;
        return *(const bf16x8*)(cA + rw * BK + (((kk * 4 + fg) ^ (rw & 7)) << 3));
      };
      auto ldw_f = [&](int kk, int i) {
        const int rw = i * 16 + fr;
        return *(const bf16x8*)(cW + rw * BK + (((kk * 4 + fg) ^ (rw & 7)) << 3));
      };
      bf16x8 af0[MI], af1[MI];
#pragma unroll
      for (int j = 0; j < MI; ++j) af0[j] = lda_f(0, j);
      if constexpr (X3M) {
#pragma unroll
        for (int j = 0; j < MI; ++j) af1[j] = lda_f(1, j);
      }
      bf16x8 wcur = ldw_f(0, 0);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const bf16x8 wnxt = (i + 1 < NI) ? ldw_f(0, i + 1) : ldw_f(1, 0);
        if (!X3M && i == NI - 1) {
#pragma unroll
          for (int j = 0; j < MI; ++j) af1[j] = lda_f(1, j);
        }
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = H16<HT>::mfma16(wcur, af0[j], acc[i][j]);
        if constexpr (X3M) {
#pragma unroll
          for (int j = 0; j < MI; ++j) acc[i][j] = H16<HT>::mfma16(wcur, af1[j], acc[i][j]);
        }
        wcur = wnxt;
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        bf16x8 wnxt = wcur;
        if (i + 1 < NI) wnxt = ldw_f(1, i + 1);
        if (i == NI - 1) lds_reads_done_barrier();  // all fragment reads retired: the stage may be refilled
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = H16<HT>::mfma16(wcur, X3M ? af0[j] : af1[j], acc[i][j]);
        wcur = wnxt;
      }
    };
    [[maybe_unused]] auto mx_tile = [&](int kt) {
      const HT *cA, *cW;
      tile_head(kt, cA, cW);
                        // M6 rows, MX tile (K / 64 is even: the parity of kt is the tile kind)
        const MxOff mo = mx_offsets(fr, fg);
        MxFrag am[MI];
#pragma unroll
        for (int j = 0; j < MI; ++j) am[j] = ld_mx_frag(cA + j * 16 * BK, mo);
        MxFrag wcur = ld_mx_frag(cW, mo);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          MxFrag wnxt = wcur;
          if (i + 1 < NI) wnxt = ld_mx_frag(cW + (i + 1) * 16 * BK, mo);
          if (i == NI - 1) lds_reads_done_barrier();
#pragma unroll
          for (int j = 0; j < MI; ++j) acc[i][j] = mfma_mx6(wcur, am[j], acc[i][j]);
          wcur = wnxt;
        }
    };
    if constexpr (MXM) {
      for (int kt = 0; kt < nk; kt += 2) { hi_tile(kt); mx_tile(kt + 1); }
    } else {
      for (int kt = 0; kt < nk; ++kt) hi_tile(kt);
    }
    g += nk;
    // LDS-staged epilogue in the stage consumed last ((g-1)&1); the other stage is receiving the next tile
    staged_epilogue<T, EPI, MI, NI>(acc, (float*)(smem + ((g - 1) & 1) * STAGE), ep, M, m0 + wm * TM, n0 + wn * TN, lane, wave);
    if (!has_next) break;
    __builtin_amdgcn_s_barrier();                   // every wave is done with its strip before the stage is refilled
    m0 = nm0; n0 = nn0; idx = nidx;
  }
}

// ------------------------------------------------------------------------------------------------
// Ping-pong persistent kernel for the large encoder GEMMs (256x256 tile, 8 waves = 2 row groups x 4 column waves,
// wave tile 128x64, K tiles of 64, two 64-KB LDS buffers of four 16-KB half-tiles [A0 | A1 | B0 | B1]).
//
// A K tile is two phases, one 128x32 half of the wave tile each (32 MFMAs).  A phase is
//     L part: fragment ds_reads + LDS-DMA prefetch issue, s_waitcnt lgkmcnt(0)      | s_barrier |
//     M part: 32 MFMAs at raised priority                                           | s_barrier |
// and row group 1 runs one barrier behind row group 0 (it takes one extra barrier at the start), so on every SIMD
// one wave is in its M part while the other is in its L part: the matrix pipe is fed by one group while the other
// issues its loads, instead of all 8 waves reading, then all 8 multiplying.
// (Rounds 2-3 ran FOUR phases of 16 MFMAs.  Cycle stamps — build --stamps 4, tools/pp_stamps.py — showed the MFMAs themselves
// at full rate, 16.5 cycles each, and every barrier interval bound by the OTHER group's L part plus the hand-over; two phases
// halve the hand-overs: qkv / o-proj / fc1 / fc2 / conv2 at 256 windows -2.4 / -2.2 / -3.0 / -5.2 / -4.9 %.)
//
// Fragment reads of K tile g:  phase A: b0, b1, a0    phase B: a1 (into a0's registers);  quadrants (a0,b0) (a0,b1) | (a1,b1) (a1,b0).
// Group x executes the L part of phase p (A = 0, B = 1) in barrier interval 2p + x and RETIRES its reads (lgkmcnt(0)) before
// it arrives at the barrier that closes that interval.  Hence, for the buffer of K tile g (intervals 4g + i), the half-tiles
// may be overwritten
//     B0, B1 (read by both groups in phase A):     from interval 4g + 2
//     A0 (read by group 0 only, phases A and B):   from interval 4g + 3
//     A1 (read by group 1 only):                   from interval 4g + 4
// and the prefetch stream is      phase B of g: B pair of K tile g+2 (intervals 4g+2 / 4g+3)
//                                 phase A of g: A half of K tile g+1 (it overwrites K tile g-1: free since 4g - 1 / 4g)
// All 8 waves share the B pair (every wave streams its eighth of both halves); the A halves are GROUP-LOCAL: the 4 waves of group x
// stream A<x>, the half only they read.  Retirement of K tile g+1, first read by group 0 in interval 4g+4 (issue order per wave:
// B pair of g+1, A half of g+1, B pair of g+2):
//     group 1, end of its phase-B L part (interval 4g+3):   s_waitcnt vmcnt(8) — its pieces of the B pair of g+1
//     both groups, behind their phase-B MFMAs (4g+3 / 4g+4): s_waitcnt vmcnt(4) — own A half (+ group 0's B pieces); the B pair of
//                                                            g+2 stays in flight across the barrier
// (Until r04 every wave streamed an eighth of all four half-tiles and retired K tile g+1 with one vmcnt(4) in its phase-B L part —
// 280-370 cycles of exposed load latency in that L part according to the stamps.)
// The K-tile stream is continuous across the output tiles a workgroup owns.  The last K tile e of an output tile
// leaves buffer e & 1 to the epilogue as its staging area: the B pair of K tile e+2 is held back to phase A of K tile
// e+1, which both groups reach only after the barrier that closes the (shared) epilogue interval.
// ------------------------------------------------------------------------------------------------
// SPLITK (decoder rows, too few 256x256 output tiles for the chip): the tile space is S copies of the m x n tile grid; copy z
// multiplies K tiles [z nk / S, (z + 1) nk / S) and stores its fp32 partial tile at out_f32 + z M N (EPI_F32, no bias); the
// caller's reduction kernel (splitk_reduce_resid_ln_kernel) sums the S planes.
template <typename T, int EPI, bool SPLITK = false>
__global__ __launch_bounds__(512) void gemm_h16_pp_kernel(const typename IO<T>::H* __restrict__ A, int lda, const typename IO<T>::H* __restrict__ W,
                                                          int ldw, int M, int N, int K, EpiParams ep, int ntm, int GM, int S) {
  typedef typename IO<T>::H HT;
  constexpr bool MXM = IsMx<T>::v;      // M6 rows: odd K tiles are MX tiles (16 scaled MFMAs per phase instead of 32 plain ones)
  constexpr bool X3M = IO<T>::split && !IsMx<T>::v;      // hi | lo K tiles: 24 instead of 16 MFMAs per phase, (W hi, A hi) (W hi, A lo) (W lo, A hi)
  constexpr bool LATEWAIT = WSEG_PP_LATEWAIT != 0;      // see hi_tile, phase B
  constexpr int BM = 256, BN = 256, BK = 64, TM = 128, TN = 64, MI = 8, NI = 4;
  constexpr int HTILE = 128 * BK;                      // elements per half-tile (16 KB)
  constexpr int BUF = 4 * HTILE;                       // elements per K-tile buffer: [A0 | A1 | B0 | B1]
  __shared__ __attribute__((aligned(16))) HT smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
#if defined(WSEG_STAMPS) && WSEG_STAMPS == 4
  // timing experiment (wrong results): GM bit 8 (WSEG_PP_SOLO) = row group 1 leaves at once, group 0 runs its L and M parts alone
  if ((GM & 0x100) && wr == 1) return;
  GM &= 0xff;
#endif
  const int fr = lane & 15, fg = lane >> 4;
  const int ntn = N / BN, ntmn = ntm * ntn, ntiles = SPLITK ? ntmn * S : ntmn, nk = K / BK;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, bpx = gridDim.x >> 3;
  const int q = ntiles >> 3, r = ntiles & 7;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  const int count = q + (xcd < r ? 1 : 0);
  if (loc >= count) return;
  // split-K: copy z of the tile grid owns K tiles [k_first(z), k_first(z + 1))
  auto tile_z = [&](int swz) { return SPLITK ? swz / ntmn : 0; };
  auto k_first = [&](int z) { return SPLITK ? (MXM ? 2 * (z * (nk / 2) / S) : z * nk / S) : 0; };      // M6 rows: whole (hi, MX) tile pairs
  int KT = ((count - loc + bpx - 1) / bpx) * nk;            // K tiles this workgroup consumes
  if constexpr (SPLITK) {
    KT = 0;
    for (int i = loc; i < count; i += bpx) { const int z = tile_z(start + i); KT += k_first(z + 1) - k_first(z); }
  }
  auto tile_coords = [&](int swz, int& m0, int& n0) {
    if constexpr (SPLITK) swz -= tile_z(swz) * ntmn;
    const int per_group = GM * ntn, grp = swz / per_group, rem = swz - grp * per_group;
    const int gm = min(GM, ntm - grp * GM);
    m0 = (grp * GM + rem % gm) * BM;
    n0 = (rem / gm) * BN;
  };

  // ---- prefetch stream: one cursor per operand (tile sequence index, K tile, stream index, scalar row base) ----
  // thread's piece of a half-tile: LDS 16-B slot p = it*512 + tid (it = 0, 1) holds row p >> 3, logical slot
  // (p & 7) ^ (row & 7) (swizzle on the source side); piece 1 is 64 rows below piece 0, same slot.
  const int prow = tid >> 3, pslot = (tid & 7) ^ (prow & 7);
  // The A half-tiles are GROUP-LOCAL (r04): row group x streams its own half-tile A<x> (its 4 waves cover the 128 rows with 4
  // pieces of 32 rows: LDS slot p = it*256 + (tid & 255), row p >> 3), so nothing a group reads from an A half depends on the other
  // group's loads — a group may retire its A pieces as late as the end of the phase-B M part before the tile is read.
  const int arow = (tid & 255) >> 3;                     // row within a 32-row piece; (arow & 7) == (prow & 7): the same slot swizzle
  // per-lane BYTE offsets, unsigned 32-bit (rows of a tile span < 2^31 bytes): with a wave-uniform 64-bit base they select the
  // SGPR-base + 32-bit-VGPR-offset form of global_load_lds — no 64-bit address arithmetic (and no 64-bit lane offsets to spill:
  // a spilled pair was reloaded behind an s_waitcnt vmcnt(0) in every phase, draining the prefetch stream)
  const unsigned a_lane = (unsigned)((wr * 128 + arow) * lda + pslot * 8) * (unsigned)sizeof(HT), w_lane = (unsigned)(prow * ldw + pslot * 8) * (unsigned)sizeof(HT);
  // The L part of a phase is a serial chain of instruction issue (r04: every VALU / SALU instruction taken out of it shortens the
  // barrier interval): the lane offsets stay in two 32-bit registers — behind an opaque copy at each use, so that the compiler
  // neither hoists a zero-extended 64-bit pair out of the loop (it spilled one in the M6 instantiation, reloaded behind a
  // vmcnt(0)) nor gives up the SGPR-base form of the instruction — and the operand streams are uniform pointers advanced by one
  // K tile per A1 / B1 issue instead of being rebuilt from the cursors.  (Recomputing the offsets from the thread index at every
  // issue, r04's first fix for the spill, cost 2.5-3 % of the f16m6 GEMMs.)
  auto lane_off = [&](unsigned kept) -> unsigned {
    unsigned t = kept;
    asm volatile("" : "+v"(t));
    return t;
  };
  int ca_idx = loc, ca_kt = 0, ca_g = 0, cb_idx = loc, cb_kt = 0, cb_g = 0;
  int tm, tn;
  tile_coords(start + loc, tm, tn);
  size_t ca_row = (size_t)tm * lda, cb_row = (size_t)tn * ldw;                    // scalar
  int ca_k0 = k_first(tile_z(start + loc)), ca_nk = SPLITK ? k_first(tile_z(start + loc) + 1) - ca_k0 : nk, cb_k0 = ca_k0, cb_nk = ca_nk;
  // uniform byte pointers of the two operand streams (row 0 of the current output tile's rows, current K tile) and the constant
  // strides of their pieces
  const size_t a_piece = (size_t)32 * lda * sizeof(HT);
  const size_t w_half = (size_t)128 * ldw * sizeof(HT), w_piece = (size_t)64 * ldw * sizeof(HT);
  constexpr size_t K_STEP = (size_t)BK * sizeof(HT);
  const char* ca_ptr = (const char*)(A + (ca_row + (size_t)(SPLITK ? ca_k0 : 0) * BK));
  const char* cb_ptr = (const char*)(W + (cb_row + (size_t)(SPLITK ? cb_k0 : 0) * BK));
  // par: LDS buffer of the K tile being issued, when the caller knows it at compile time (M6 rows: a K range is whole (hi, MX) tile
  // pairs, so hi tiles always live in buffer 0 and MX tiles in buffer 1 — the buffer selects and the fragment address arithmetic fold
  // away); -1: the stream index decides
  auto issue_a = [&](int par = -1) {                   // this row group's half A<wr> of K tile ca_g; the cursor advances
    HT* dst = smem + (par >= 0 ? par : (ca_g & 1)) * BUF + wr * HTILE + wc * 512;
    const unsigned al = lane_off(a_lane);
    WSEG_GLDS16(ca_ptr + (size_t)al, dst);
    WSEG_GLDS16(ca_ptr + a_piece + (size_t)al, dst + 2048);
    WSEG_GLDS16(ca_ptr + 2 * a_piece + (size_t)al, dst + 4096);
    WSEG_GLDS16(ca_ptr + 3 * a_piece + (size_t)al, dst + 6144);
    ++ca_g;
    ca_ptr += K_STEP;
    if (++ca_kt == (SPLITK ? ca_nk : nk)) {
      ca_kt = 0;
      ca_idx += bpx;
      if (ca_idx < count) {
        tile_coords(start + ca_idx, tm, tn);
        ca_row = (size_t)tm * lda;
        if constexpr (SPLITK) { const int z = tile_z(start + ca_idx); ca_k0 = k_first(z); ca_nk = k_first(z + 1) - ca_k0; }
        ca_ptr = (const char*)(A + (ca_row + (size_t)(SPLITK ? ca_k0 : 0) * BK));
      }
    }
  };
  auto issue_b = [&](int par = -1) {
    HT* dst = smem + (par >= 0 ? par : (cb_g & 1)) * BUF + 2 * HTILE + wave * 512;
    const unsigned wl = lane_off(w_lane);
    WSEG_GLDS16(cb_ptr + (size_t)wl, dst);
    WSEG_GLDS16(cb_ptr + w_piece + (size_t)wl, dst + 4096);
    WSEG_GLDS16(cb_ptr + w_half + (size_t)wl, dst + HTILE);
    WSEG_GLDS16(cb_ptr + w_half + w_piece + (size_t)wl, dst + HTILE + 4096);
    ++cb_g;
    cb_ptr += K_STEP;
    if (++cb_kt == (SPLITK ? cb_nk : nk)) {
      cb_kt = 0;
      cb_idx += bpx;
      if (cb_idx < count) {
        tile_coords(start + cb_idx, tm, tn);
        cb_row = (size_t)tn * ldw;
        if constexpr (SPLITK) { const int z = tile_z(start + cb_idx); cb_k0 = k_first(z); cb_nk = k_first(z + 1) - cb_k0; }
        cb_ptr = (const char*)(W + (cb_row + (size_t)(SPLITK ? cb_k0 : 0) * BK));
      }
    }
  };

  // prologue: K tile 0 complete, B pair of K tile 1 in flight
  issue_b(); issue_a();
  if (KT > 1) { issue_b(); wait_vmcnt<4>(); }
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one barrier behind group 0

  // fragment addressing: row j*16 + fr of a half-tile, 16-B slot (kk*4 + fg) ^ (fr & 7); kk = 1 is the kk = 0 address ^ 64 B
  const int frag0 = fr * BK + ((fg ^ (fr & 7)) << 3);
  const int fa0 = wr * HTILE + frag0, fa1 = fa0 ^ 32;
  const int fb0 = (2 + (wc >> 1)) * HTILE + (wc & 1) * 64 * BK + frag0, fb1 = fb0 ^ 32;
  [[maybe_unused]] const MxOff mxo = mx_offsets(fr, fg);

#define WSEG_PP_QUADRANT(JA, IB)                                                                                      \
  _Pragma("unroll") for (int kk = 0; kk < (X3M ? 3 : 2); ++kk)                                                        \
    _Pragma("unroll") for (int i = 2 * (IB); i < 2 * (IB) + 2; ++i)                                                   \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
        acc[i][4 * (JA) + j] = H16<HT>::mfma16(bfr[i][X3M ? (kk >> 1) : kk], afr[j][X3M ? (kk & 1) : kk], acc[i][4 * (JA) + j]);
  // closes an L part (this wave's fragment reads have RETURNED before it signals: the stage may be refilled behind this barrier)
  // and runs the M part of two quadrants
#define WSEG_PP_MFMA(JA, IB0, IB1)                                                                                    \
  do {                                                                                                                \
    WSEG_PP_STAMP(pp_phase, 1);                                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                \
    __builtin_amdgcn_s_barrier();                                                                                     \
    WSEG_PP_STAMP(pp_phase, 2);                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                                    \
    WSEG_PP_QUADRANT(JA, IB0)                                                                                         \
    WSEG_PP_QUADRANT(JA, IB1)                                                                                         \
    __builtin_amdgcn_s_setprio(0);                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    WSEG_PP_STAMP(pp_phase, 3);                                                                                       \
  } while (0)

  int g = 0;                                           // K tiles consumed so far
  f32x4 acc[NI][MI];
  for (int idx = loc; idx < count; idx += bpx) {
  int m0, n0;
  tile_coords(start + idx, m0, n0);
  const int tz = tile_z(start + idx), nkt = SPLITK ? k_first(tz + 1) - k_first(tz) : nk;
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // One K tile = two phases.  The tile bodies are lambdas so that the M6 instantiation can run a (hi tile, MX tile) PAIR per loop
  // iteration as straight-line code: with both bodies behind a branch inside one loop the register allocator spilled ~60 VGPRs into
  // the loop, and every scratch reload waits (vmcnt is in order) for the whole LDS-DMA prefetch stream: 4x slower than f16x3.
  // MID (a compile-time tag): a K tile in the middle of an output tile — not its first, not its last, and at least two more K tiles
  // to come: every prefetch condition is known, the L parts lose their compares and branches (the L part is a serial chain of
  // instruction issue; r04: ~1 % of the GEMM per 8-10 instructions taken out of it)
  auto hi_tile = [&](int kt, auto mid_tag) {
    constexpr bool MID = decltype(mid_tag)::value;
    const bool first = !MID && kt == 0 && g > 0, final = !MID && kt == nkt - 1;
    constexpr int PAR = MXM ? 0 : -1;                  // this tile's LDS buffer (M6 rows: hi tiles in buffer 0), -1: g & 1
    constexpr int NPAR = MXM ? 1 : -1;
    const HT* cur = smem + (MXM ? 0 : (g & 1)) * BUF;
    bf16x8 afr[4][2], bfr[NI][2];
    // ---- phase A: b0, b1, a0 -> quadrants (a0, b0), (a0, b1); A pair of K tile g+1 ----
    [[maybe_unused]] int pp_phase = 0;
    WSEG_PP_STAMP(0, 0);
    WSEG_PP_CLOCK();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bfr[i][0] = *(const bf16x8*)(cur + i * 16 * BK + fb0);
      bfr[i][1] = *(const bf16x8*)(cur + i * 16 * BK + fb1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      afr[j][0] = *(const bf16x8*)(cur + j * 16 * BK + fa0);
      afr[j][1] = *(const bf16x8*)(cur + j * 16 * BK + fa1);
    }
    if (first && g + 1 < KT) issue_b(NPAR);                     // B pair of K tile g+1, held back over the epilogue
    if (MID || g + 1 < KT) issue_a(NPAR);                       // A pair of K tile g+1
    __builtin_amdgcn_sched_barrier(0);
    WSEG_PP_MFMA(0, 0, 1);
    __builtin_amdgcn_s_barrier();
    // ---- phase B: a1 -> quadrants (a1, b1), (a1, b0); B pair of K tile g+2, retire K tile g+1 ----
    pp_phase = 1;
    WSEG_PP_STAMP(1, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      afr[j][0] = *(const bf16x8*)(cur + (4 + j) * 16 * BK + fa0);
      afr[j][1] = *(const bf16x8*)(cur + (4 + j) * 16 * BK + fa1);
    }
    // Retirement of K tile g+1 (LATEWAIT, r04; the stamps showed 280 cycles of exposed load latency in this L part): group 1's pieces
    // of its B pair — group 0 reads them at the start of the next interval but one — are retired here, everything else (a group's
    // own A half, group 0's B pieces) only behind this phase's MFMAs, one interval later.  Issue order per wave: B pair of g+1,
    // A half of g+1, B pair of g+2.
    const bool pb = MID || (!final && g + 2 < KT);
    if (pb) issue_b(PAR);                                           // B pair of K tile g+2
    if constexpr (LATEWAIT) {
      if (wr == 1) { if (pb) wait_vmcnt<8>(); else wait_vmcnt<4>(); }
    } else {
      if (pb) wait_vmcnt<4>(); else wait_vmcnt<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    WSEG_PP_MFMA(1, 1, 0);
    if constexpr (LATEWAIT) { if (pb) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
    if (!final) __builtin_amdgcn_s_barrier();
  };
  [[maybe_unused]] auto mx_tile = [&](int kt, auto mid_tag) {
    constexpr bool MID = decltype(mid_tag)::value;
    const bool first = !MID && kt == 0 && g > 0, final = !MID && kt == nkt - 1;
    const HT* cur = smem + BUF;                        // MX tiles live in buffer 1
      // ---- MX tile (M6 rows): the same two phases, hand-overs and prefetch stream as below; the fragments are 24-byte e2m3
      // groups + a scale byte (ld_mx_frag), a quadrant pair is 16 scaled MFMAs ----
      const unsigned aT = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)(cur + wr * HTILE);      // this row group's 128-row A half-tile
      const unsigned bT = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)(cur + (2 + (wc >> 1)) * HTILE + (wc & 1) * 64 * BK);   // this wave's 64 W rows
      MxFrag7 bm[NI], am[4];
      WSEG_PP_STAMP(0, 0);
      const lds_cp a0 = (lds_cp)(uintptr_t)(aT + (unsigned)mxo.p0), a1 = (lds_cp)(uintptr_t)(aT + (unsigned)mxo.p1);
      const lds_cp b0 = (lds_cp)(uintptr_t)(bT + (unsigned)mxo.p0), b1 = (lds_cp)(uintptr_t)(bT + (unsigned)mxo.p1);
      bm[0] = ld_mx_frag7<0>(b0, b1); bm[1] = ld_mx_frag7<2048>(b0, b1); bm[2] = ld_mx_frag7<4096>(b0, b1); bm[3] = ld_mx_frag7<6144>(b0, b1);
      am[0] = ld_mx_frag7<0>(a0, a1); am[1] = ld_mx_frag7<2048>(a0, a1); am[2] = ld_mx_frag7<4096>(a0, a1); am[3] = ld_mx_frag7<6144>(a0, a1);
      if (first && g + 1 < KT) issue_b(0);
      if (MID || g + 1 < KT) issue_a(0);
      __builtin_amdgcn_sched_barrier(0);
      WSEG_PP_STAMP(0, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      WSEG_PP_STAMP(0, 2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma_mx6_asm(acc[i][j], bm[i], am[j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      WSEG_PP_STAMP(0, 3);
      __builtin_amdgcn_s_barrier();
      WSEG_PP_STAMP(1, 0);
      am[0] = ld_mx_frag7<8192>(a0, a1); am[1] = ld_mx_frag7<10240>(a0, a1); am[2] = ld_mx_frag7<12288>(a0, a1); am[3] = ld_mx_frag7<14336>(a0, a1);
      const bool pb = MID || (!final && g + 2 < KT);
      if (pb) issue_b(1);
      if constexpr (LATEWAIT) {
        if (wr == 1) { if (pb) wait_vmcnt<8>(); else wait_vmcnt<4>(); }
      } else {
        if (pb) wait_vmcnt<4>(); else wait_vmcnt<0>();
      }
      __builtin_amdgcn_sched_barrier(0);
      WSEG_PP_STAMP(1, 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      WSEG_PP_STAMP(1, 2);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mfma_mx6_asm(acc[i][4 + j], bm[i], am[j]);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (LATEWAIT) { if (pb) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
      WSEG_PP_STAMP(1, 3);
      if (!final) __builtin_amdgcn_s_barrier();
  };
  constexpr std::false_type EDGE{};
  constexpr std::true_type MIDDLE{};
  // three stretches of straight-line code (one loop body behind a branch spilled hundreds of bytes): leading edge tile(s), the
  // middle tiles, trailing edge tiles
  if constexpr (MXM) {      // K ranges are whole (hi, MX) tile pairs starting at a hi tile
    int kt = 0;
    hi_tile(kt, EDGE); ++g; mx_tile(kt + 1, EDGE); ++g;
    kt = 2;
#pragma nounroll
    for (; kt + 3 < nkt; kt += 2) { hi_tile(kt, MIDDLE); ++g; mx_tile(kt + 1, MIDDLE); ++g; }
#pragma nounroll
    for (; kt < nkt; kt += 2) { hi_tile(kt, EDGE); ++g; mx_tile(kt + 1, EDGE); ++g; }
  } else {
    int kt = 0;
    hi_tile(kt, EDGE); ++g;
    kt = 1;
#pragma nounroll
    for (; kt + 2 < nkt; ++kt, ++g) hi_tile(kt, MIDDLE);
#pragma nounroll
    for (; kt < nkt; ++kt, ++g) hi_tile(kt, EDGE);
  }
  // Epilogue: group 0 gives up its one-barrier lead (it idles while group 1 finishes its last 16 MFMAs), both groups run
  // their epilogues in the SAME interval — back to back they cost two epilogue times with the matrix pipe idle, side by
  // side the loads / stores of 8 waves overlap (o-proj 459 -> 441 us, fc1 1653 -> 1598 us at 256 windows) — and group 1
  // then drops one barrier behind again.  Every wave stages in its own strip of the buffer the last K tile left; nothing
  // is prefetched into that buffer before phase A of the next K tile, which both groups reach only after the barrier
  // below.  (Requesting group 0's residual rows before its idle interval measured no further gain.)
  {
    if (wr == 0) __builtin_amdgcn_s_barrier();
    if constexpr (SPLITK) {
      EpiParams epz = ep;
      epz.out_f32 = ep.out_f32 + (size_t)tz * M * N;
      staged_epilogue<T, EPI, MI, NI>(acc, (float*)(smem + ((g - 1) & 1) * BUF), epz, M, m0 + wr * TM, n0 + wc * TN, lane, wave);
    } else {
      staged_epilogue<T, EPI, MI, NI>(acc, (float*)(smem + ((g - 1) & 1) * BUF), ep, M, m0 + wr * TM, n0 + wc * TN, lane, wave);
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();
  }
  }
#undef WSEG_PP_MFMA
#undef WSEG_PP_QUADRANT
  if (wr == 0) __builtin_amdgcn_s_barrier();          // pair group 1's extra barrier
}


template <int EPI, typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int m_pad,
                                                            int M, int N, EpiParams ep) {
  const int nq = N >> 2;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * nq) return;
  const int m = idx / nq, n0 = (idx - m * nq) << 2;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < splits; ++z) {
    const float4 t = *(const float4*)(part + ((size_t)z * m_pad + m) * N + n0);
    v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
  }
  epi_apply<EPI, T>(ep, m, n0, v);
}

// The same with 8 columns per thread through the 8-column epilogue: in WSEG_F16M6 mode (EPI_STORE / EPI_GELU, N % 32 == 0) the four
// lanes of a quad cover one 32-column block and write M6 rows directly — the split-K skinny path then hands the next GEMM its operand
// without a conversion launch (small batches: one graph node per decoder layer less).  Same summation order as the 4-column kernel.
template <int EPI, typename T>
__global__ __launch_bounds__(256) void splitk_reduce8_kernel(const float* __restrict__ part, int splits, int m_pad,
                                                             int M, int N, EpiParams ep) {
  const int n8 = N >> 3;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * n8) return;      // M * n8 is a multiple of 4 (N % 32 == 0): whole quads leave together
  const int m = idx / n8, n0 = (idx - m * n8) << 3;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < splits; ++z) {
    const float* p = part + ((size_t)z * m_pad + m) * N + n0;
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  epi_apply8<EPI, T>(ep, m, n0, v);
}

// ------------------------------------------------------------------------------------------------
// Live profiler of the dominant kernel (wseg_profile_begin / wseg_profile_end)
// ------------------------------------------------------------------------------------------------
// Process-wide and meant for ONE measuring thread (bench.py's roofline leg); a mutex keeps concurrent device threads from
// corrupting the event pool if a profile is requested while several devices run.
struct GemmProfiler {
  std::mutex mu;
  bool on = false;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  std::vector<double> flops;
  hipEvent_t get() {
    if (used == pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool.push_back(e); }
    return pool[used++];
  }
};
static GemmProfiler g_prof;

// ------------------------------------------------------------------------------------------------
// Launchers
// ------------------------------------------------------------------------------------------------
// CU count of the CURRENT device, cached per device (thread-per-device mode drives several devices from one process).
int device_cu_count() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

#define WSEG_TRY_(expr) do { int _s = (expr); if (_s != WSEG_OK) return _s; } while (0)

// skinny family (decoder steps, small encoders): BN = 64, BM in {32, 64, 128}; K is split across workgroups until
// ~256+ of them stream the weight matrix.  Partials are reduced (in a fixed order) by a second kernel that
// applies the epilogue (or the fused residual + LayerNorm).
struct SkinnyPlan { int bm, bn, mt, m_pad, splits, k_len; };

// The rows a launch is PLANNED for (GemmArgs::plan_m): kernel family, tile size and split-K ranges follow them, the grid follows g.M.
static inline int plan_rows(const GemmArgs& g) { return g.plan_m > 0 ? g.plan_m : g.M; }
static inline GemmArgs plan_view(const GemmArgs& g) { GemmArgs p = g; p.M = plan_rows(g); p.plan_m = 0; return p; }

static SkinnyPlan plan_skinny(const GemmArgs& g0, bool pairs = false) {      // pairs: M6 rows — a K range is whole (hi, MX) tile pairs
  const GemmArgs g = plan_view(g0);
  SkinnyPlan sp;
  // largest row tile that still yields >= 160 workgroups without splitting K; otherwise 128 rows + split-K
  sp.bm = g.M <= 32 ? 32 : (g.M <= 64 ? 64 : 128);
  static const int force_bm = WSEG_KNOB_INT("WSEG_SKINNY_BM", 0);   // tuning knob (variant builds)
  if (g.M > 64) {
    const int nt = g.N / 64;
    if (nt * cdiv(g.M, 128) < 160 && nt * cdiv(g.M, 64) >= 160) sp.bm = 64;
    if (force_bm == 64 || force_bm == 128) sp.bm = force_bm;
  }
  sp.bn = 64;
  sp.mt = cdiv(g.M, sp.bm);
  sp.m_pad = sp.mt * sp.bm;
  const int blocks = (g.N / sp.bn) * sp.mt;
  sp.splits = 1;
  if (g.splitk_ws) {
    // split K (in whole 64-wide tiles, >= 2 tiles per split) until ~256 workgroups stream the weights
    const int nk = g.K / 64;
    static const int target = WSEG_KNOB_INT("WSEG_SKINNY_TARGET", 256);   // tuning knob (variant builds)
    for (int cand = 2; cand <= 16 && blocks * sp.splits < target; ++cand) {
      if (nk % cand || nk / cand < 2 || (pairs && (nk / cand) % 2)) continue;
      if ((size_t)cand * sp.m_pad * g.N * sizeof(float) > g.splitk_ws_bytes) break;
      sp.splits = cand;
    }
  }
  sp.k_len = g.K / sp.splits;
  sp.mt = cdiv(g0.M, sp.bm);      // the grid and the partial planes cover the rows actually launched
  sp.m_pad = sp.mt * sp.bm;
  return sp;
}

// fp32 partial sums [splits][m_pad][N] into g.splitk_ws (valid for splits == 1 too)
template <typename T>
static int launch_skinny_partial(const GemmArgs& g, const SkinnyPlan& sp, hipStream_t s) {
  typedef typename IO<T>::H HT;
  const HT* A = (const HT*)g.A;
  const HT* W = (const HT*)g.W;
  if (!g.splitk_ws || (size_t)sp.splits * sp.m_pad * g.N * sizeof(float) > g.splitk_ws_bytes) {
    set_error("split-K workspace missing or too small");
    return WSEG_ERR_STATE;
  }
  dim3 grid(g.N / sp.bn, sp.mt, sp.splits);
  // LDS ring depth: 2 stages.  Measured (large, 120 windows): 6-8 K tiles in flight with one workgroup per CU is 1.7x
  // SLOWER than 3-4 stages at two workgroups per CU, and 2 stages (3-5 workgroups per CU) is another 2-3 % faster at every
  // batch size — these kernels want co-resident workgroups to cover their barriers, not more bytes in flight each.
  // Round 2, at 1024 rows (profiles/README.md, "decode GEMM experiments"): 3 / 4 stages 12.8 / 16.7 us against 11.1 us;
  // 128x128 tiles (split 4) 13.6 us; no split with 128x128 tiles and 3 stages 24 us; register staging (global -> VGPR ->
  // ds_write, one barrier per K tile) 35.8 us; dropping the loads OR the MFMAs from the loop changes nothing (11.3 / 9.8
  // us): a K tile costs ~0.6 us of serialised issue -> land -> barrier -> fragment reads -> MFMA -> barrier per workgroup,
  // while a bare LDS-DMA stream of the same L2-resident data runs at 123 GB/s per CU (tools/probes/l2fill_probe.hip).
#define WSEG_SKINNY_P(BM_, WM_, WN_)                                                                                     \
  hipLaunchKernelGGL((gemm_h16_kernel<T, BM_, 64, WM_, WN_, EPI_STORE, true, 2>), grid, dim3(256), 0, s, A, g.lda, W, g.ldw, g.M, \
                     g.N, sp.k_len, g.ep, g.splitk_ws, sp.m_pad, 0)
  if (sp.bm == 32) WSEG_SKINNY_P(32, 1, 4);
  else if (sp.bm == 64) WSEG_SKINNY_P(64, 1, 4);
  else WSEG_SKINNY_P(128, 2, 2);
#undef WSEG_SKINNY_P
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

// x[m][:] += bias + sum_z part[z][m][:]  (x is the fp32 residual stream), then y[m][:] = LayerNorm(x[m][:]) in the model dtype.
// One workgroup per row, one 8-element chunk per thread (d <= 2048), all split partials loaded up front.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_resid_ln_kernel(const float* __restrict__ part, int splits, int m_pad, int M,
                                                                     int d, const typename IO<T>::P* __restrict__ bias, float* __restrict__ x,
                                                                     const typename IO<T>::P* __restrict__ gam, const typename IO<T>::P* __restrict__ bet,
                                                                     void* __restrict__ y) {
  typedef typename IO<T>::P PT;
  __shared__ float s_red[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = tid * 8;
  const bool act = c < d;
  float v[8];
  float sum = 0.f;
  if (act) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* pp = part + (size_t)row * d + c;
    const size_t zs = (size_t)m_pad * d;
#pragma unroll 4
    for (int z = 0; z < splits; ++z) {
      const float4 p0 = *(const float4*)(pp + z * zs), p1 = *(const float4*)(pp + z * zs + 4);
      a[0] += p0.x; a[1] += p0.y; a[2] += p0.z; a[3] += p0.w; a[4] += p1.x; a[5] += p1.y; a[6] += p1.z; a[7] += p1.w;
    }
    float* xp = x + (size_t)row * d + c;
    const float4 x0 = *(const float4*)xp, x1 = *(const float4*)(xp + 4);
    const float r[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    float bb[8];
    ld8_h<PT>(bias + c, bb);
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = r[j] + (a[j] + bb[j]); sum += v[j]; }
    *(float4*)xp = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(xp + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  sum = wave_sum(sum);
  if (lane == 0) s_red[wave] = sum;
  __syncthreads();
  const float mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)d;
  __syncthreads();
  float sq = 0.f;
  if (act) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float t = v[j] - mean; sq += t * t; }
  }
  sq = wave_sum(sq);
  if (lane == 0) s_red[wave] = sq;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)d + 1e-5f);
  if (act) {
    float gg[8], be[8], o[8];
    ld8_h<PT>(gam + c, gg);
    ld8_h<PT>(bet + c, be);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (v[j] - mean) * rstd * gg[j] + be[j];
    op_st8<T>(y, (size_t)row, d, c, o);
  }
}

// Large-tile kernels (128x128 / 256x256 output tiles, no split-K) or the weight-stream family (64-column tiles, split-K)?
// Decoder-step GEMMs sit at the border.  Measured on the whole decode step (large, 4 beams): at 1024 rows fc1 (N = 5120,
// 320 blocks of 128x128) is better in the stream family (decode step 10.71 -> 10.23 ms); at 1536 rows q|k|v (360 blocks)
// and fc1 (480) are better on large tiles (13.50 vs 13.70 ms).
static bool big_tile_path(const GemmArgs& g) {
  static const long big_min = WSEG_KNOB_INT("WSEG_BIG_MIN_BLOCKS", 340);   // tuning knob (variant builds)
  const int pm = plan_rows(g);
  return pm > 128 && g.N % 128 == 0 && (long)cdiv(pm, 128) * (g.N / 128) >= big_min;
}

// Split-precision modes: the caller's K / lda / ldw are LOGICAL; the kernels see rows of 2K 16-bit words.
template <typename T> static GemmArgs kernel_view(const GemmArgs& g0) {
  GemmArgs g = g0;
  if (IO<T>::split) { g.K *= 2; g.lda *= 2; g.ldw *= 2; }
  return g;
}

template <typename T> static int launch_pp_splitk(const GemmArgs& g, int S, hipStream_t s);
template <typename T> static int pp_splitk_plan(const GemmArgs& g);

static bool skinny_split_writes_mx(int N) {
  static const bool off = WSEG_KNOB_SET("WSEG_NO_MX_REDUCE");      // A/B knob (variant builds)
  return !off && N % 32 == 0;
}

template <int EPI, typename T>
static int launch_h16(const GemmArgs& g0, hipStream_t s) {
  typedef typename IO<T>::H HT;
  const GemmArgs g = kernel_view<T>(g0);
  const HT* A = (const HT*)g.A;
  const HT* W = (const HT*)g.W;
  if (g.K % 64 || g.N % 64 || (IsMx<T>::v && g.K % 128)) { set_error("gemm h16: K %d / N %d not tile multiples", g.K, g.N); return WSEG_ERR_INVALID; }
  if (big_tile_path(g)) {
    static const bool no_swz = WSEG_KNOB_SET("WSEG_NO_XCD_SWIZZLE");
    static const bool big256 = !WSEG_KNOB_SET("WSEG_GEMM_128");   // 256x256 tiles by default where they fill the chip
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g_prof.on) {
      std::lock_guard<std::mutex> lk(g_prof.mu);
      e0 = g_prof.get(); e1 = g_prof.get(); g_prof.flops.push_back(2.0 * g0.M * g0.N * g0.K); (void)hipEventRecord(e0, s);
    }
    static const bool persist = !WSEG_KNOB_SET("WSEG_GEMM_NO_PERSIST");
    // m-tiles per tile group of the persistent order: the 32 tiles in flight on one XCD then span ~4 activation tiles x 8
    // weight tiles, the smallest operand footprint for 32 tiles (a + b = 12 operand tiles; 2-row groups re-stream the
    // whole weight matrix per tile pair: PMC FETCH_SIZE 2-3x the algorithmic bytes, profiles/).  Measured at 256
    // windows: 4 beats 2 by 2-8 % on the K = 1280 shapes, 6 and 8 lose on K = 5120.
#if defined(WSEG_STAMPS) && WSEG_STAMPS == 4
    static const int group_m = WSEG_KNOB_INT("WSEG_GEMM_GROUP_M", 4) | (getenv("WSEG_PP_SOLO") ? 0x100 : 0);
#else
    static const int group_m = WSEG_KNOB_INT("WSEG_GEMM_GROUP_M", 4);
#endif
    const int n_cu = device_cu_count();
    // 256x256 tiles need whole rounds of the chip: with fewer than 4 rounds, a last round that leaves more than a fifth of
    // the CUs idle costs more than the smaller tile's lower arithmetic intensity (decoder fc1 at 4096 rows: 320 tiles =
    // 1.25 rounds, 118 us against 1280 tiles of 128x128 in 2.5 rounds of 512 workgroups).
    const int pm = plan_rows(g);
    const long nt256 = (long)cdiv(pm, 256) * (g.N / 256);
    const long rounds256 = (nt256 + n_cu - 1) / n_cu;
    static const bool quant_rule = !WSEG_KNOB_SET("WSEG_GEMM_NO_QUANT_RULE");   // tuning knob (variant builds)
    // (split / mixed modes: a K tile pair costs twice the bf16 K tile while the 128x128 kernel's fixed costs do not shrink — since the
    // r04 K-loop work the 256x256 kernel wins down to 3/5 of a last round: decoder fc1 at 4 096 rows, 320 tiles, 152 against 161 us)
    const bool ragged256 = quant_rule && rounds256 < 4 && nt256 * 5 < rounds256 * n_cu * (IO<T>::split ? 3 : 4);
    // (r05: 160 tiles for the GELU epilogue in the split / mixed modes — decoder fc1 at 2 048 rows, one round on 62 % of the CUs: 512 slots 21.3 ->
    // 20.3 ms per decode step; NOT for the q|k|v epilogue: 180 tiles at 3 072 rows lose to two rounds of 128x128 tiles, 28.0 -> 29.0 ms)
    static const int min_tiles256 = WSEG_KNOB_INT("WSEG_BIG256_MIN_TILES", 0);
    const long need256 = min_tiles256 ? min_tiles256 : ((IO<T>::split && EPI == EPI_GELU) ? 160 : 192);
    if (big256 && g.N % 256 == 0 && nt256 >= need256 && !ragged256 && g.K >= 128) {      // (K = 64 words: one K tile, 128x128 kernel)
      const int ntm = cdiv(g.M, 256), ntiles = ntm * (g.N / 256);
      constexpr bool pingpong = true;      // (r05: the generic / persistent kernels' 256x256 instantiations — knob-only paths that spilled 200-330 bytes per lane — are gone)
      // WSEG_F16M6, M6-row outputs, between one and two rounds of 256x256 tiles (decoder fc1 at 4 096 rows: 320 tiles on 256 CUs — the
      // second round runs on a quarter of the chip): the columns that fill ONE round go through the kernel as usual; the remaining
      // column tiles are multiplied as split-K copies that fill the chip once more for 1 / S of the K range, and the 8-column
      // reduction writes their M6 rows.
      if constexpr (IsMx<T>::v && (EPI == EPI_STORE || EPI == EPI_GELU)) {
        static const bool tail_split = !WSEG_KNOB_SET("WSEG_NO_TAIL_SPLIT");      // A/B knob (variant builds)
        const int ptm = cdiv(pm, 256);      // planned row tiles: the column split and S must not follow the rows launched
        const int ntn = g.N / 256, full_cols = n_cu / ptm, pairs = g.K / 128;
        if (tail_split && pingpong && rounds256 == 2 && g.splitk_ws && full_cols >= 1 && full_cols < ntn && g.K >= 256) {
          const int rem_tiles = (ntn - full_cols) * ptm, n1 = full_cols * 256, n2 = g.N - n1;
          int S = n_cu / rem_tiles;
          while (S >= 2 && ((rem_tiles * S) % 8 || pairs / S < 2 || (size_t)S * pm * n2 * sizeof(float) > g.splitk_ws_bytes)) --S;
          if (S >= 2 && rem_tiles * S * 4 >= n_cu * 3) {
            int grid = ntm * full_cols < n_cu ? ntm * full_cols : n_cu;
            grid = (grid + 7) & ~7;      // (a multiple of 8: one share per XCD; workgroups beyond the tile count return at once)
            hipLaunchKernelGGL((gemm_h16_pp_kernel<T, EPI>), dim3(grid), dim3(512), 0, s, A, g.lda, W, g.ldw, g.M, n1, g.K, g.ep, ntm, group_m, 1);
            GemmArgs g2 = g;
            g2.W = W + (size_t)n1 * g.ldw;
            g2.N = n2;
            WSEG_TRY_(launch_pp_splitk<T>(g2, S, s));
            typedef typename IO<T>::P PT;
            EpiParams e3 = g.ep;
            if (e3.bias) e3.bias = (const PT*)g.ep.bias + n1;
            e3.out = (char*)g.ep.out + (size_t)(n1 >> 6) * 256;      // M6 rows: 256 bytes per 64 logical columns; ldc stays the full row
            hipLaunchKernelGGL((splitk_reduce8_kernel<EPI, T>), dim3(cdiv(g.M * (n2 / 8), 256)), dim3(256), 0, s, g.splitk_ws, S, g.M, g.M, n2, e3);
            if (e1) (void)hipEventRecord(e1, s);
            WSEG_LAUNCH_CHECK();
            return WSEG_OK;
          }
        }
      }
      // (Measured and dropped: r04, the generic kernel as 4 waves x 128x128 wave tiles, one wave per SIMD — 640 against 1 080 TFLOP/s on the
      // encoder shapes; r05, the hand-built one-wave-per-SIMD kernel — tools/experiments/r05_gemm_w4_kernel.hip.txt, profiles/r05_w4_experiments.txt;
      // r03, a register-resident residual epilogue — all CUs reach their epilogue together and its 1.3 GB of fp32 residual traffic is an
      // HBM-bound burst either way: staged 540 / 1513 us, direct 590 / 1525 us; r04, hi-only attention projections in f16m6 — 19x the logit
      // error at 32 layers.)
      {
        int grid = ntiles < n_cu ? ntiles : n_cu;
        grid = (grid + 7) & ~7;      // (a multiple of 8: one share per XCD; workgroups beyond the tile count return at once)
        hipLaunchKernelGGL((gemm_h16_pp_kernel<T, EPI>), dim3(grid), dim3(512), 0, s, A, g.lda, W, g.ldw, g.M, g.N, g.K, g.ep, ntm, group_m, 1);
      }
    } else {
      const int ntm = cdiv(g.M, 128), ntiles = ntm * (g.N / 128);
      if (persist && !no_swz && cdiv(pm, 128) * (g.N / 128) >= 16) {
        int grid = ntiles < 2 * n_cu ? ntiles : 2 * n_cu;
        grid = (grid + 7) & ~7;      // (a multiple of 8: one share per XCD; workgroups beyond the tile count return at once)
        hipLaunchKernelGGL((gemm_h16_persist_kernel<T, 128, 128, 2, 2, EPI>), dim3(grid), dim3(256), 0, s, A, g.lda, W, g.ldw, g.M, g.N,
                           g.K, g.ep, ntm, group_m);
      } else {
        dim3 grid(g.N / 128, ntm, 1);
        if (!no_swz) grid = dim3(ntiles, 1, 1);
        hipLaunchKernelGGL((gemm_h16_kernel<T, 128, 128, 2, 2, EPI, false>), grid, dim3(256), 0, s, A, g.lda, W, g.ldw, g.M, g.N,
                           g.K, g.ep, (float*)nullptr, 0, no_swz ? 0 : ntm);
      }
    }
    if (e1) (void)hipEventRecord(e1, s);
    WSEG_LAUNCH_CHECK();
    return WSEG_OK;
  }
#ifndef WSEG_MX_PP_SPLITK
#define WSEG_MX_PP_SPLITK 1
#endif
  // WSEG_F16M6, M6-row outputs below the large-tile threshold but with hundreds of rows (decoder fc1 at 112-270 slots; r05): split-K copies
  // of the 256x256 kernel + the M6-writing 8-column reduction, instead of the 128x64 stream kernel (whose 4-column epilogue writes hi | lo
  // rows that a conversion launch turns into M6 rows): 57.5 + 11.8 us -> see profiles/r05_epilogue_ab.txt.  gemm_out_is_mx predicts it.
  if constexpr (IsMx<T>::v && (EPI == EPI_STORE || EPI == EPI_GELU)) {
    const int S = WSEG_MX_PP_SPLITK ? pp_splitk_plan<T>(g) : 0;
    if (S) {
      WSEG_TRY_(launch_pp_splitk<T>(g, S, s));
      hipLaunchKernelGGL((splitk_reduce8_kernel<EPI, T>), dim3(cdiv(g.M * (g.N / 8), 256)), dim3(256), 0, s, g.splitk_ws, S, g.M, g.M, g.N, g.ep);
      WSEG_LAUNCH_CHECK();
      return WSEG_OK;
    }
  }
  SkinnyPlan sp = plan_skinny(g, IsMx<T>::v);
  // (block-floating-point cross K / V: the row writer needs the lanes of a row side by side, which the 4-column MFMA-layout epilogue of
  // the stream kernels does not give — an un-split plan goes through the partial plane + reduction kernel as well)
  const bool coop_kv = EPI == EPI_KV_CROSS && IO<T>::split && g.ep.kv24 >= 2;
  if (sp.splits == 1 && !(coop_kv && g.splitk_ws && (size_t)sp.m_pad * g.N * sizeof(float) <= g.splitk_ws_bytes)) {
    if (coop_kv) { set_error("gemm: block-floating-point cross K / V needs the split-K workspace"); return WSEG_ERR_STATE; }
    dim3 grid(g.N / 64, sp.mt, 1);
#define WSEG_SKINNY(BM_, WM_, WN_)                                                                                      \
  hipLaunchKernelGGL((gemm_h16_kernel<T, BM_, 64, WM_, WN_, EPI, false, 2>), grid, dim3(256), 0, s, A, g.lda, W, g.ldw, g.M, g.N, \
                     g.K, g.ep, (float*)nullptr, sp.m_pad, 0)
    if (sp.bm == 32) WSEG_SKINNY(32, 1, 4);
    else if (sp.bm == 64) WSEG_SKINNY(64, 1, 4);
    else WSEG_SKINNY(128, 2, 2);
#undef WSEG_SKINNY
    WSEG_LAUNCH_CHECK();
    return WSEG_OK;
  }
  WSEG_TRY_(launch_skinny_partial<T>(g, sp, s));
  if constexpr (IsMx<T>::v && (EPI == EPI_STORE || EPI == EPI_GELU)) {
    if (skinny_split_writes_mx(g.N)) {      // M6 rows straight from the reduction (gemm_out_is_mx predicts exactly this)
      hipLaunchKernelGGL((splitk_reduce8_kernel<EPI, T>), dim3(cdiv(g.M * (g.N / 8), 256)), dim3(256), 0, s, g.splitk_ws, sp.splits, sp.m_pad, g.M,
                         g.N, g.ep);
      WSEG_LAUNCH_CHECK();
      return WSEG_OK;
    }
  }
  const int work = g.M * (g.N / 4);
  hipLaunchKernelGGL((splitk_reduce_kernel<EPI, T>), dim3(cdiv(work, 256)), dim3(256), 0, s, g.splitk_ws, sp.splits, sp.m_pad, g.M, g.N, g.ep);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

template <int EPI>
static int launch_any(int dtype, const GemmArgs& g, hipStream_t s) {
  if (dtype == WSEG_BF16) return launch_h16<EPI, bf16_t>(g, s);
  if (dtype == WSEG_F16) return launch_h16<EPI, f16_t>(g, s);
  if (dtype == WSEG_BF16X3) return launch_h16<EPI, X3<bf16_t>>(g, s);
  if (dtype == WSEG_F16X3) return launch_h16<EPI, X3<f16_t>>(g, s);
  if (dtype == WSEG_F16M6) return launch_h16<EPI, M6>(g, s);
  return launch_gemm_f32((EpiKind)EPI, g, s);      // exact-parity kernels: wseg_gemm_f32.hip
}

template <typename T>
static int gemm_partial_t(const GemmArgs& g0, PartialInfo* info, bool* ok, hipStream_t s) {
  const GemmArgs g = kernel_view<T>(g0);
#ifndef WSEG_PARTIAL_PP
#define WSEG_PARTIAL_PP 1
#endif
  static const bool partial_big = !WSEG_KNOB_SET("WSEG_NO_PARTIAL_BIG");
  // (r05: also ABOVE the large-tile threshold when the 256x256 tiles fill at most half the chip — decoder q|k|v at 2 048 rows, 120 tiles: two split-K
  // copies + the reduction inside the attention kernel instead of the 128x128 kernel with its own epilogue)
  if (!g.splitk_ws || g.K % 64 || g.N % 64) return WSEG_OK;
  if (big_tile_path(g) && !(WSEG_PARTIAL_PP && partial_big && pp_splitk_plan<T>(g) >= 2)) return WSEG_OK;
  if (WSEG_PARTIAL_PP) {      // thousands of rows, too few 256x256 tiles for the chip (the decode step's cross-attention query): the split-K copies
    const int S = pp_splitk_plan<T>(g);      // of the ping-pong kernel leave the same fp32 planes [z][M][N] as the stream family
    if (S) {
      WSEG_TRY_(launch_pp_splitk<T>(g, S, s));
      info->part = g.splitk_ws; info->splits = S; info->m_pad = g.M; info->n = g.N;
      *ok = true;
      return WSEG_OK;
    }
  }
  SkinnyPlan sp = plan_skinny(g, IsMx<T>::v);
  if ((size_t)sp.splits * sp.m_pad * g.N * sizeof(float) > g.splitk_ws_bytes) return WSEG_OK;
  WSEG_TRY_(launch_skinny_partial<T>(g, sp, s));
  info->part = g.splitk_ws; info->splits = sp.splits; info->m_pad = sp.m_pad; info->n = g.N;
  *ok = true;
  return WSEG_OK;
}

int launch_gemm_partial(int dtype, const GemmArgs& g, PartialInfo* info, bool* ok, hipStream_t s) {
  *ok = false;
  switch (dtype) {
    case WSEG_BF16: return gemm_partial_t<bf16_t>(g, info, ok, s);
    case WSEG_F16: return gemm_partial_t<f16_t>(g, info, ok, s);
    case WSEG_BF16X3: return gemm_partial_t<X3<bf16_t>>(g, info, ok, s);
    case WSEG_F16X3: return gemm_partial_t<X3<f16_t>>(g, info, ok, s);
    case WSEG_F16M6: return gemm_partial_t<M6>(g, info, ok, s);
    default: return WSEG_OK;      // exact-parity mode: no split-K (one k-ordered chain per output)
  }
}

// x = x + (A W^T + bias); y = LayerNorm(x) * gamma + beta.   MFMA decoder rows: split-K partials + ONE fused
// reduction/residual/LayerNorm kernel; otherwise the generic GEMM (EPI_RESID) followed by launch_layernorm.
// Split-K on the 256x256 ping-pong kernel for decoder rows with a long K and too few output tiles for the chip (fc2 at 4096
// rows: 16 x 5 = 80 tiles on 256 CUs; the 128x64 stream kernel needs 99 us for it, hipBLASLt 56): S = n_cu / tiles copies of the
// tile grid, each multiplying 1/S of the K tiles into an fp32 partial plane.  Returns 0 when the shape does not qualify.
template <typename T> static int pp_splitk_plan(const GemmArgs& g) {
  // tuning knobs.  Split / mixed modes (twice the K tiles per logical column, r04 K loop): the split-K 256x256 kernel beats the
  // 128x64 stream kernel from 512 rows up (decode step at 128 / 256 / 384 slots: 9.9 -> 9.3, 15.2 -> 14.5, 18.4 -> 17.7 ms; at 64
  // slots the stream kernel wins, 6.2 against 6.7 ms)
  static const int min_rows_env = WSEG_KNOB_INT("WSEG_PP_SPLITK_MIN_ROWS", 0);
  const int min_rows = min_rows_env ? min_rows_env : (IO<T>::split ? 448 : 2048);      // (r05: 480 rows = the 120 windows of a one-hour recording, 9.39 -> 8.99 ms per step; at 384 rows the stream kernel wins, 7.32 against 7.77)
  static const int min_kt = WSEG_KNOB_INT("WSEG_PP_SPLITK_MIN_KT", 40);
  const int pm = plan_rows(g);
  if (pm < min_rows || g.N % 256 || g.K % 64 || g.K / 64 < min_kt || !g.splitk_ws) return 0;
  const int nt = cdiv(pm, 256) * (g.N / 256), nk = g.K / 64, n_cu = device_cu_count();
  static const int max_s = WSEG_KNOB_INT("WSEG_PP_SPLITK_MAX_S", 64), min_kt_per = WSEG_KNOB_INT("WSEG_PP_SPLITK_KT_PER", 4);
  int S = n_cu / nt;
  if (S > max_s) S = max_s;
  while (S >= 2 && ((nt * S) % 8 || nk / S < min_kt_per || (size_t)S * pm * g.N * sizeof(float) > g.splitk_ws_bytes)) --S;
  // fewer than 96 workgroups of 5 K tiles each (d x d projections at 448-511 rows) lose to the stream family: 120 windows 8.9-9.1 -> 8.6 ms per
  // decode step, 112 windows 8.73 -> 8.13 (profiles/r05_epilogue_ab.txt)
  static const int min_wgs = WSEG_KNOB_INT("WSEG_PP_SPLITK_MIN_WGS", 96);
  if (nt * S < min_wgs) return 0;
  return S >= 2 ? S : 0;
}

template <typename T> static int launch_pp_splitk(const GemmArgs& g, int S, hipStream_t s) {
  typedef typename IO<T>::H HT;
  static const int group_m = WSEG_KNOB_INT("WSEG_GEMM_GROUP_M", 4);
  const int ntm = cdiv(g.M, 256), ntiles = ntm * (g.N / 256) * S, n_cu = device_cu_count();
  int grid = ntiles < n_cu ? ntiles : n_cu;
  grid = (grid + 7) & ~7;      // (a multiple of 8: one share per XCD; workgroups beyond the tile count return at once)
  EpiParams ep;
  ep.out_f32 = g.splitk_ws; ep.ldc = g.N;
  hipLaunchKernelGGL((gemm_h16_pp_kernel<T, EPI_F32, true>), dim3(grid), dim3(512), 0, s, (const HT*)g.A, g.lda, (const HT*)g.W, g.ldw,
                     g.M, g.N, g.K, ep, ntm, group_m, S);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

template <typename T>
static int gemm_resid_ln_t(const GemmArgs& g0, const void* gamma, const void* beta, void* y, bool* done, hipStream_t s) {
  typedef typename IO<T>::P PT;
  const GemmArgs g = kernel_view<T>(g0);
  const int d = g.N;
  if (d % 8 == 0 && d <= 2048 && g.ep.bias && g.ep.resid == g.ep.out && g.ep.ldc == d && !big_tile_path(g)) {
    const int S = pp_splitk_plan<T>(g);
    if (S) {
      WSEG_TRY_(launch_pp_splitk<T>(g, S, s));
      hipLaunchKernelGGL(splitk_reduce_resid_ln_kernel<T>, dim3(g.M), dim3(256), 0, s, g.splitk_ws, S, g.M, g.M, d, (const PT*)g.ep.bias,
                         (float*)g.ep.out, (const PT*)gamma, (const PT*)beta, y);
      WSEG_LAUNCH_CHECK();
      *done = true;
      return WSEG_OK;
    }
  }
  if (big_tile_path(g) || !g.splitk_ws || d % 8 || d > 2048 || !g.ep.bias || g.ep.resid != g.ep.out || g.ep.ldc != d || g.K % 64 ||
      g.N % 64)
    return WSEG_OK;
  SkinnyPlan sp = plan_skinny(g, IsMx<T>::v);
  static const bool fuse_unsplit = !WSEG_KNOB_SET("WSEG_RESID_LN_ALWAYS_PARTIAL");   // tuning knob (variant builds)
  // K not split (enough row tiles to fill the chip, 2048+ rows): the fp32 partial round trip buys nothing; the GEMM adds
  // the residual in its own epilogue and a LayerNorm launch follows (2048 rows: 21.6 + ~6 us against 26.7 + 8.9 us)
  if (!(sp.splits > 1 || !fuse_unsplit) || (size_t)sp.splits * sp.m_pad * g.N * sizeof(float) > g.splitk_ws_bytes) return WSEG_OK;
  WSEG_TRY_(launch_skinny_partial<T>(g, sp, s));
  hipLaunchKernelGGL(splitk_reduce_resid_ln_kernel<T>, dim3(g.M), dim3(256), 0, s, g.splitk_ws, sp.splits, sp.m_pad, g.M, d,
                     (const PT*)g.ep.bias, (float*)g.ep.out, (const PT*)gamma, (const PT*)beta, y);
  WSEG_LAUNCH_CHECK();
  *done = true;
  return WSEG_OK;
}

int launch_gemm_resid_ln(int dtype, const GemmArgs& g, const void* gamma, const void* beta, void* y, hipStream_t s) {
  bool done = false;
  switch (dtype) {
    case WSEG_BF16: WSEG_TRY_(gemm_resid_ln_t<bf16_t>(g, gamma, beta, y, &done, s)); break;
    case WSEG_F16: WSEG_TRY_(gemm_resid_ln_t<f16_t>(g, gamma, beta, y, &done, s)); break;
    case WSEG_BF16X3: WSEG_TRY_(gemm_resid_ln_t<X3<bf16_t>>(g, gamma, beta, y, &done, s)); break;
    case WSEG_F16X3: WSEG_TRY_(gemm_resid_ln_t<X3<f16_t>>(g, gamma, beta, y, &done, s)); break;
    case WSEG_F16M6: WSEG_TRY_(gemm_resid_ln_t<M6>(g, gamma, beta, y, &done, s)); break;
    default: break;
  }
  if (done) return WSEG_OK;
  WSEG_TRY_(launch_gemm(dtype, EPI_RESID, g, s));
  return launch_layernorm(dtype, (const float*)g.ep.out, gamma, beta, y, g.M, g.N, s);      // (WSEG_F16M6: y as M6 rows, like the fused kernel)
}

// WSEG_F16M6: are the operand rows an EPI_STORE / EPI_GELU launch of this shape writes M6 rows (LDS-staged 8-column epilogues of the
// large-tile kernels: cooperative op_st8) or hi | lo rows (4-column epilogues of the skinny family and its split-K reduction)?
bool gemm_out_is_mx(int dtype, int M, int N, int K, size_t splitk_ws_bytes) {
  if (dtype != WSEG_F16M6) return false;
  GemmArgs g;
  g.M = M; g.N = N; g.K = 2 * K;
  if (big_tile_path(g)) return true;
  if (!splitk_ws_bytes || g.K % 128 || N % 64) return false;
  g.splitk_ws = (float*)(uintptr_t)16;      // any non-null value: the plans only ask whether a workspace exists and how large it is
  g.splitk_ws_bytes = splitk_ws_bytes;
  if (WSEG_MX_PP_SPLITK && pp_splitk_plan<M6>(g)) return true;
  return plan_skinny(g, true).splits > 1 && skinny_split_writes_mx(N);
}

int launch_gemm(int dtype, EpiKind epi, const GemmArgs& g, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return launch_any<EPI_STORE>(dtype, g, s);
    case EPI_GELU: return launch_any<EPI_GELU>(dtype, g, s);
    case EPI_RESID: return launch_any<EPI_RESID>(dtype, g, s);
    case EPI_GELU_POS: return launch_any<EPI_GELU_POS>(dtype, g, s);
    case EPI_QKV_ENC: return launch_any<EPI_QKV_ENC>(dtype, g, s);
    case EPI_KV_CROSS: return launch_any<EPI_KV_CROSS>(dtype, g, s);
    case EPI_F32: return launch_any<EPI_F32>(dtype, g, s);
    case EPI_QKV_DEC: return launch_any<EPI_QKV_DEC>(dtype, g, s);
    case EPI_SCALE: return launch_any<EPI_SCALE>(dtype, g, s);
    default: break;
  }
  set_error("unknown epilogue %d", (int)epi);
  return WSEG_ERR_INVALID;
}

}  // namespace wseg

extern "C" int wseg_debug_gemm(int32_t dtype, int32_t epi, int32_t M, int32_t N, int32_t K, const void* A, const void* W,
                               const void* bias, const void* resid, void* out, void* splitk_ws, size_t splitk_ws_bytes,
                               void* stream) {
  using namespace wseg;
  if (!A || !W || !out || M <= 0 || N <= 0 || K <= 0 || epi < 0 || epi > 2) { set_error("wseg_debug_gemm: bad argument"); return WSEG_ERR_INVALID; }
  GemmArgs g;
  g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K;
  g.ep.bias = bias; g.ep.out = out; g.ep.ldc = N; g.ep.resid = resid;
  g.splitk_ws = (float*)splitk_ws; g.splitk_ws_bytes = splitk_ws_bytes;
  return launch_gemm(dtype, epi == 0 ? EPI_STORE : (epi == 1 ? EPI_GELU : EPI_RESID), g, (hipStream_t)stream);
}

extern "C" int wseg_debug_gemm_out_is_mx(int32_t dtype, int32_t M, int32_t N, int32_t K) {
  return wseg::gemm_out_is_mx(dtype, M, N, K, (size_t)1 << 40) ? 1 : 0;      // wseg_debug_gemm with a workspace that never limits the split
}

extern "C" int wseg_debug_gemm_resid_ln(int32_t dtype, int32_t M, int32_t N, int32_t K, const void* A, const void* W, const void* bias,
                                        void* x, const void* gamma, const void* beta, void* y, void* splitk_ws, size_t splitk_ws_bytes,
                                        void* stream) {
  using namespace wseg;
  if (!A || !W || !bias || !x || !gamma || !beta || !y || M <= 0 || N <= 0 || K <= 0) { set_error("wseg_debug_gemm_resid_ln: bad argument"); return WSEG_ERR_INVALID; }
  GemmArgs g;
  g.A = A; g.lda = K; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K;
  g.ep.bias = bias; g.ep.out = x; g.ep.resid = x; g.ep.ldc = N;
  g.splitk_ws = (float*)splitk_ws; g.splitk_ws_bytes = splitk_ws_bytes;
  return launch_gemm_resid_ln(dtype, g, gamma, beta, y, (hipStream_t)stream);
}

extern "C" int wseg_profile_begin(void) {
  using namespace wseg;
  g_prof.on = true;
  g_prof.used = 0;
  g_prof.flops.clear();
  return WSEG_OK;
}

extern "C" int wseg_profile_end(double* total_flops, double* total_ms, int64_t* launches) {
  using namespace wseg;
  if (!total_flops || !total_ms || !launches) { set_error("wseg_profile_end: null argument"); return WSEG_ERR_INVALID; }
  g_prof.on = false;
  double fl = 0.0, ms = 0.0;
  for (size_t i = 0; i < g_prof.flops.size(); ++i) {
    WSEG_HIP_CHECK(hipEventSynchronize(g_prof.pool[2 * i + 1]));
    float t = 0.f;
    WSEG_HIP_CHECK(hipEventElapsedTime(&t, g_prof.pool[2 * i], g_prof.pool[2 * i + 1]));
    ms += t;
    fl += g_prof.flops[i];
  }
  *total_flops = fl; *total_ms = ms; *launches = (int64_t)g_prof.flops.size();
  return WSEG_OK;
}

#if defined(WSEG_STAMPS) && WSEG_STAMPS == 6 && defined(WSEG_KNOBS)
extern "C" int wseg_debug_w4_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wseg::g_w4_stamps), sizeof(unsigned long long) * (4 * 512 + 32));
}
#endif
#if defined(WSEG_STAMPS) && WSEG_STAMPS == 4
extern "C" int wseg_debug_pp_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wseg::g_pp_stamps), sizeof(unsigned long long) * 132);
}
#endif
