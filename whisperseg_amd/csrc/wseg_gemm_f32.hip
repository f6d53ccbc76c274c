// Exact-parity fp32 GEMM kernels (WSEG_F32 mode) for gfx950:  C[m][n] = sum_k A[m][k] * W[n][k]  with the fused epilogues of wseg_gemm_epi.h.
// Every variant computes each dot product as ONE k-ordered fmaf chain (v_mfma_f32_32x32x2_f32 / 16x16x4 are such chains bit for bit), so
// all of them — and whatever tile plan the launcher picks — give the same bits (tests/test_gemm_gpu.py; test knob WSEG_F32_GEMM).
#include <stdlib.h>
#include <string.h>
#include "wseg_gemm_epi.h"

namespace wseg {

// ------------------------------------------------------------------------------------------------
// f32 exact kernel: 64x64 tile, 4x4 micro-tile per thread, sequential-k fmaf chain.
// ------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                       int M, int N, int K, EpiParams ep) {
  __shared__ float sA[16][68];
  __shared__ float sW[16][68];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int lr = tid >> 2, lq = (tid & 3) * 4;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float4 av = *(const float4*)(A + (size_t)(m0 + lr) * lda + k0 + lq);
    const float4 wv = *(const float4*)(W + (size_t)(n0 + lr) * ldw + k0 + lq);
    sA[lq + 0][lr] = av.x; sA[lq + 1][lr] = av.y; sA[lq + 2][lr] = av.z; sA[lq + 3][lr] = av.w;
    sW[lq + 0][lr] = wv.x; sW[lq + 1][lr] = wv.y; sW[lq + 2][lr] = wv.z; sW[lq + 3][lr] = wv.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], w[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = sA[k][ty * 4 + i]; w[i] = sW[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m < M) epi_apply<EPI, float>(ep, m, n0 + tx * 4, acc[i]);
  }
}

// f32 exact kernel for large problems: 128x128 tile, 8x8 micro-tile per thread (twice the FMAs per LDS float of the 64x64
// kernel, which is LDS-bound at ~30 % of the fp32 VALU peak), next K slab prefetched into registers under the FMAs.  Every
// output element is still ONE fmaf chain over k = 0, 1, 2, ... — bit-identical to gemm_f32_kernel, whatever the tiling.
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_big_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                              int M, int N, int K, EpiParams ep) {
  constexpr int BK = 16, LD = 128 + 4;
  __shared__ float sA[2][BK][LD];
  __shared__ float sW[2][BK][LD];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  // loads: 128 rows x 16 k per operand = 512 float4, two per thread: rows lr and lr + 64, k offset lq
  const int lr = tid >> 2, lq = (tid & 3) * 4;
  const float* ap = A + (size_t)(m0 + lr) * lda + lq;
  const float* wp = W + (size_t)(n0 + lr) * ldw + lq;
  const size_t a64 = (size_t)64 * lda, w64 = (size_t)64 * ldw;
  float4 ra0 = *(const float4*)ap, ra1 = *(const float4*)(ap + a64);
  float4 rw0 = *(const float4*)wp, rw1 = *(const float4*)(wp + w64);
  auto stage = [&](int b) {
    sA[b][lq + 0][lr] = ra0.x; sA[b][lq + 1][lr] = ra0.y; sA[b][lq + 2][lr] = ra0.z; sA[b][lq + 3][lr] = ra0.w;
    sA[b][lq + 0][lr + 64] = ra1.x; sA[b][lq + 1][lr + 64] = ra1.y; sA[b][lq + 2][lr + 64] = ra1.z; sA[b][lq + 3][lr + 64] = ra1.w;
    sW[b][lq + 0][lr] = rw0.x; sW[b][lq + 1][lr] = rw0.y; sW[b][lq + 2][lr] = rw0.z; sW[b][lq + 3][lr] = rw0.w;
    sW[b][lq + 0][lr + 64] = rw1.x; sW[b][lq + 1][lr + 64] = rw1.y; sW[b][lq + 2][lr + 64] = rw1.z; sW[b][lq + 3][lr + 64] = rw1.w;
  };
  stage(0);
  __syncthreads();
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 acc2[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc2[i][j] = (f2){0.f, 0.f};
  const int nk = K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int b = kt & 1;
    if (kt + 1 < nk) {
      ra0 = *(const float4*)(ap + (kt + 1) * BK); ra1 = *(const float4*)(ap + a64 + (kt + 1) * BK);
      rw0 = *(const float4*)(wp + (kt + 1) * BK); rw1 = *(const float4*)(wp + w64 + (kt + 1) * BK);
    }
#pragma unroll 2
    for (int k = 0; k < BK; ++k) {          // (a full unroll hoists all 64 LDS reads and spills)
      // rows ty*4 + {0..3} and 64 + ty*4 + {0..3}; columns tx*4 + {0..3} and 64 + tx*4 + {0..3}: 16-byte LDS reads
      const float4 a0 = *(const float4*)&sA[b][k][ty * 4], a1 = *(const float4*)&sA[b][k][64 + ty * 4];
      const float4 w0 = *(const float4*)&sW[b][k][tx * 4], w1 = *(const float4*)&sW[b][k][64 + tx * 4];
      const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const f2 w2[4] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}};
      // two columns per v_pk_fma_f32 (each half is an IEEE fma: same bits as fmaf)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f2 aa = {a[i], a[i]};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc2[i][j] = __builtin_elementwise_fma(aa, w2[j], acc2[i][j]);
      }
    }
    if (kt + 1 < nk) stage(b ^ 1);          // the other buffer was last read before the barrier that closed slab kt - 1
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + (i >> 2) * 64 + ty * 4 + (i & 3);
    if (m < M) {
      float v0[4] = {acc2[i][0][0], acc2[i][0][1], acc2[i][1][0], acc2[i][1][1]};
      float v1[4] = {acc2[i][2][0], acc2[i][2][1], acc2[i][3][0], acc2[i][3][1]};
      epi_apply<EPI, float>(ep, m, n0 + tx * 4, v0);
      epi_apply<EPI, float>(ep, m, n0 + 64 + tx * 4, v1);
    }
  }
}

// f32 exact kernels on the fp32 matrix cores: v_mfma_f32_32x32x2_f32 is bit for bit a k-ordered fmaf chain (one rounding per
// product, no wider accumulation; MI355X_MICROARCH.md), so these produce exactly the bits of gemm_f32_kernel at the f32 VECTOR
// rate but with two LDS dwords per 4096 FMAs instead of one per two.  2 x 2 waves, wave tile (32 TI) x (32 TJ); the weight
// rows are the MFMA's A operand and the activation rows its B operand, so a lane ends up with 4 consecutive output columns.
template <int EPI, int TI, int TJ>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                            int M, int N, int K, EpiParams ep) {
  constexpr int BM = 64 * TI, BN = 64 * TJ, BK = 16, LDA = BM + 4, LDW = BN + 4;
  __shared__ float sA[2][BK][LDA];
  __shared__ float sW[2][BK][LDW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int lr = tid >> 2, lq = (tid & 3) * 4;          // global loads: row lr (+ 64 i), k offset lq, 16 bytes each
  const float* ap = A + (size_t)(m0 + lr) * lda + lq;
  const float* wp = W + (size_t)(n0 + lr) * ldw + lq;
  float4 ra[TI], rw[TJ];
  auto fetch = [&](int kt) {
#pragma unroll
    for (int i = 0; i < TI; ++i) ra[i] = *(const float4*)(ap + (size_t)(64 * i) * lda + kt * BK);
#pragma unroll
    for (int j = 0; j < TJ; ++j) rw[j] = *(const float4*)(wp + (size_t)(64 * j) * ldw + kt * BK);
  };
  auto stage = [&](int b) {
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      sA[b][lq + 0][lr + 64 * i] = ra[i].x; sA[b][lq + 1][lr + 64 * i] = ra[i].y;
      sA[b][lq + 2][lr + 64 * i] = ra[i].z; sA[b][lq + 3][lr + 64 * i] = ra[i].w;
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      sW[b][lq + 0][lr + 64 * j] = rw[j].x; sW[b][lq + 1][lr + 64 * j] = rw[j].y;
      sW[b][lq + 2][lr + 64 * j] = rw[j].z; sW[b][lq + 3][lr + 64 * j] = rw[j].w;
    }
  };
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  fetch(0);
  stage(0);
  __syncthreads();
  const int nk = K / BK, fi = lane & 31, fk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    const int b = kt & 1;
    if (kt + 1 < nk) fetch(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {                 // lanes 0-31 carry k = kk, lanes 32-63 k = kk + 1: ascending k
      float af[TI], wf[TJ];
#pragma unroll
      for (int i = 0; i < TI; ++i) af[i] = sA[b][kk + fk][wm * 32 * TI + i * 32 + fi];
#pragma unroll
      for (int j = 0; j < TJ; ++j) wf[j] = sW[b][kk + fk][wn * 32 * TJ + j * 32 + fi];
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) stage(b ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TI; ++i) {
    const int m = m0 + wm * 32 * TI + i * 32 + fi;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        epi_apply<EPI, float>(ep, m, n0 + wn * 32 * TJ + j * 32 + 8 * q + 4 * fk, v);
      }
  }
}

// The same for problems whose 64x64 tiles would not fill the chip (decoder steps, small encoders): 32x32 tile, 2 x 2 waves of
// one v_mfma_f32_16x16x4_f32 accumulator each (also a k-ordered fmaf chain: same bits again).
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_mfma16_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                              int M, int N, int K, EpiParams ep) {
  constexpr int BK = 16, LD = 32 + 4;
  __shared__ float sA[2][BK][LD];
  __shared__ float sW[2][BK][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  // loads: 32 rows x 16 k per operand = 128 float4: threads 0-127 fetch A, 128-255 fetch W
  const bool isw = tid >= 128;
  const int lt = tid & 127, lr = lt >> 2, lq = (lt & 3) * 4;
  const float* src = isw ? W + (size_t)(n0 + lr) * ldw + lq : A + (size_t)(m0 + lr) * lda + lq;
  float4 rg = *(const float4*)src;
  auto stage = [&](int b) {
    float (*dst)[LD] = isw ? sW[b] : sA[b];
    dst[lq + 0][lr] = rg.x; dst[lq + 1][lr] = rg.y; dst[lq + 2][lr] = rg.z; dst[lq + 3][lr] = rg.w;
  };
  stage(0);
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int nk = K / BK, fi = lane & 15, fk = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int b = kt & 1;
    if (kt + 1 < nk) rg = *(const float4*)(src + (kt + 1) * BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4)                   // lane group fk carries k = kk + fk: ascending k inside the instruction
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(sW[b][kk + fk][wn * 16 + fi], sA[b][kk + fk][wm * 16 + fi], acc, 0, 0, 0);
    if (kt + 1 < nk) stage(b ^ 1);
    __syncthreads();
  }
  const int m = m0 + wm * 16 + fi;
  if (m < M) {
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    epi_apply<EPI, float>(ep, m, n0 + wn * 16 + 4 * fk, v);
  }
}


template <int EPI>
static int launch_f32(const GemmArgs& g, hipStream_t s) {
  if (g.K % 16 || g.N % 64) { set_error("gemm f32: K %d / N %d not tile multiples", g.K, g.N); return WSEG_ERR_INVALID; }
  // WSEG_F32_GEMM = valu64 | valu128 | mfma64 | mfma (default): test knob; all four compute the same bits
  static const char* mode_env = getenv("WSEG_F32_GEMM");
  static const int mode = !mode_env ? 3 : (!strcmp(mode_env, "valu64") ? 0 : !strcmp(mode_env, "valu128") ? 1 : !strcmp(mode_env, "mfma64") ? 2 : 3);
  const bool big = g.N % 128 == 0 && (long)cdiv(g.M, 128) * (g.N / 128) >= 2L * device_cu_count();
  if (mode == 1 && big) {
    dim3 gridb(g.N / 128, cdiv(g.M, 128));
    hipLaunchKernelGGL((gemm_f32_big_kernel<EPI>), gridb, dim3(256), 0, s, (const float*)g.A, g.lda, (const float*)g.W, g.ldw, g.M, g.N, g.K, g.ep);
    WSEG_LAUNCH_CHECK();
    return WSEG_OK;
  }
  if (mode >= 2) {
    if (mode == 3 && big) {
      dim3 gridb(g.N / 128, cdiv(g.M, 128));
      hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, 2, 2>), gridb, dim3(256), 0, s, (const float*)g.A, g.lda, (const float*)g.W, g.ldw, g.M, g.N, g.K, g.ep);
    } else if (mode == 3 && (long)cdiv(g.M, 64) * (g.N / 64) < device_cu_count()) {
      dim3 gridt(g.N / 32, cdiv(g.M, 32));
      hipLaunchKernelGGL((gemm_f32_mfma16_kernel<EPI>), gridt, dim3(256), 0, s, (const float*)g.A, g.lda, (const float*)g.W, g.ldw, g.M, g.N, g.K, g.ep);
    } else {
      dim3 grids(g.N / 64, cdiv(g.M, 64));
      hipLaunchKernelGGL((gemm_f32_mfma_kernel<EPI, 1, 1>), grids, dim3(256), 0, s, (const float*)g.A, g.lda, (const float*)g.W, g.ldw, g.M, g.N, g.K, g.ep);
    }
    WSEG_LAUNCH_CHECK();
    return WSEG_OK;
  }
  dim3 grid(g.N / 64, cdiv(g.M, 64));
  hipLaunchKernelGGL((gemm_f32_kernel<EPI>), grid, dim3(256), 0, s, (const float*)g.A, g.lda, (const float*)g.W, g.ldw, g.M, g.N, g.K, g.ep);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}

int launch_gemm_f32(EpiKind epi, const GemmArgs& g, hipStream_t s) {
  switch (epi) {
    case EPI_STORE: return launch_f32<EPI_STORE>(g, s);
    case EPI_GELU: return launch_f32<EPI_GELU>(g, s);
    case EPI_RESID: return launch_f32<EPI_RESID>(g, s);
    case EPI_GELU_POS: return launch_f32<EPI_GELU_POS>(g, s);
    case EPI_QKV_ENC: return launch_f32<EPI_QKV_ENC>(g, s);
    case EPI_KV_CROSS: return launch_f32<EPI_KV_CROSS>(g, s);
    case EPI_F32: return launch_f32<EPI_F32>(g, s);
    case EPI_QKV_DEC: return launch_f32<EPI_QKV_DEC>(g, s);
    case EPI_SCALE: return launch_f32<EPI_SCALE>(g, s);
    default: break;
  }
  set_error("unknown epilogue %d", (int)epi);
  return WSEG_ERR_INVALID;
}

}  // namespace wseg
