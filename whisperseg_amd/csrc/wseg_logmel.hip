// Log-mel front-end for gfx950: framed STFT (periodic Hann, reflect-centred) -> |X|^2 -> slaney mel
// -> log10 -> per-window (max - 8) clamp -> (x + 4) / 4, for every window of a recording at once.
//
// Math follows HF feature_extraction_whisper.py:105-133 / audio_utils.py:809-1017 as configured by
// reference audio_utils.py:45-76; window slicing, zero padding, truncation to total_spec_columns and
// the min-fill follow reference model.py:138-161.
//
// Bound: HBM (algorithmic bytes per window = 4*win_len in + 4*n_mels*n_cols out); the FFT itself is
// ~0.05 GFLOP/window.  Two STFT kernels write log10-mel values straight into the output image (no raw scratch) and the
// window maximum / minimum into two ordered-uint words; a second pass clamps and normalises in place.
//   logmel_fft_kernel<256 | 512> (n_fft = 512 / 1024: every 16 / 32 / 48 kHz configuration of the reference): one WAVE per frame, the 256-point
//     packed complex FFT as four radix-4 DIF stages on 4 points per lane — butterflies and twiddles in registers (packed
//     fp32 math), three transposes through a 2-KB wave-private LDS buffer whose XOR swizzle makes every one of them
//     bank-conflict free, no workgroup barrier until the 32 frames of a workgroup are written out as 128-byte rows.
//   logmel_stft_kernel (any n_fft up to 8192): one workgroup = FPB consecutive frames, radix-2 in LDS with the twiddle table
//     staged in LDS and a __syncthreads per stage (barrier-bound: 8 % of the HBM roof; kept for the rare wide FFTs).
// Both project onto the mel filters through their contiguous non-zero bin ranges (sparse triangles, <= 2 filters per bin).
#include <stdlib.h>
#include "wseg_common.h"

namespace wseg {

__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

constexpr int LM_FB = 32;          // frames per workgroup of the fast path (one 128-byte output row segment per filter)
constexpr int LM_CH = 8;           // bins per mel work item
constexpr int LM_MAX_ITEMS = 256;  // mel work items (filters cut into <= LM_CH-bin chunks; <= (2 * 513 + 8 * 96) / 8 = 225 for 96 filters over 513 bins)
// Mel work items of one filterbank: filter m = items cb[m] .. cb[m+1]-1, item n = bins k0[n] .. k0[n]+7 with zero-padded weights.
// Built once per call by logmel_items_kernel (a few microseconds), copied into LDS by every workgroup of the fast path.
struct LmTables {
  int n_items;
  int cb[97];
  int k0[LM_MAX_ITEMS];
  float w[LM_MAX_ITEMS][LM_CH];
};

struct LogmelArgs {
  wseg_logmel_desc d;
  const float* audio;
  int64_t n_audio;
  const int64_t* win_start;
  int64_t win_len;
  int32_t n_frames;   // floor(win_len / hop): frames kept after HF drops the last one
  int32_t fpb;        // frames per workgroup
  int32_t lg_nc;      // log2(n_fft / 2)
  const struct LmTables* tables;   // mel work items of this filterbank (logmel_items_kernel), fast path only
  float* out;         // [W][n_mels][n_cols]: log10 mel of the first n_cols frames (normalised in place by the finish kernel)
  uint32_t* stats;    // [W][2] ordered-uint (max over all frames, min over the first n_cols frames)
};

extern __shared__ __attribute__((aligned(16))) unsigned char lm_smem[];

__global__ __launch_bounds__(256) void logmel_stft_kernel(LogmelArgs a) {
  const int n_fft = a.d.n_fft, nc = n_fft >> 1, lg = a.lg_nc, fpb = a.fpb;
  float2* tw = (float2*)lm_smem;                 // [nc]      e^{-2 pi i k / n_fft}
  float2* z = tw + nc;                           // [fpb][nc] packed frames / FFT in place
  float* pw = (float*)(z + (size_t)fpb * nc);    // [fpb][nc + 1] power spectrum
  const int tid = threadIdx.x;
  const int w = blockIdx.y;
  const int f0 = blockIdx.x * fpb;
  const int64_t wstart = a.win_start[w];
  const int64_t L = a.win_len;

  for (int k = tid; k < nc; k += 256) tw[k] = ((const float2*)a.d.twiddle)[k];

  // A. load, window, pack z[n] = x[2n] + i x[2n+1] at the bit-reversed slot.
  for (int idx = tid; idx < fpb * nc; idx += 256) {
    const int fr = idx / nc, n = idx - fr * nc;
    const int f = f0 + fr;
    float2 v = make_float2(0.f, 0.f);
    if (f < a.n_frames) {
      float xs[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int j = 2 * n + q;
        int64_t jl = (int64_t)f * a.d.hop - (n_fft >> 1) + j;        // window-local sample
        if (jl < 0) jl = -jl;                                        // reflect (np.pad mode="reflect")
        if (jl >= L) jl = 2 * (L - 1) - jl;
        const int64_t ai = wstart + jl;
        const float s = (ai >= 0 && ai < a.n_audio) ? a.audio[ai] : 0.f;
        xs[q] = s * a.d.window[j];
      }
      v = make_float2(xs[0], xs[1]);
    }
    const int r = (int)(__brev((unsigned)n) >> (32 - lg));
    z[fr * nc + r] = v;
  }
  __syncthreads();

  // B. in-place radix-2 DIT over nc complex points per frame.
  const int half_nc = nc >> 1;
  for (int s = 0; s < lg; ++s) {
    const int half = 1 << s;
    for (int idx = tid; idx < fpb * half_nc; idx += 256) {
      const int fr = idx / half_nc, j = idx - fr * half_nc;
      const int pos = j & (half - 1);
      const int i0 = ((j >> s) << (s + 1)) + pos;
      const int i1 = i0 + half;
      const float2 wv = tw[(pos << (lg - s - 1)) * 2];   // e^{-2 pi i pos / (2 half)} = tw_nfft[2 * pos * nc / (2 half)]
      float2* zf = z + fr * nc;
      const float2 u = zf[i0], t = zf[i1];
      const float2 v = make_float2(t.x * wv.x - t.y * wv.y, t.x * wv.y + t.y * wv.x);
      zf[i0] = make_float2(u.x + v.x, u.y + v.y);
      zf[i1] = make_float2(u.x - v.x, u.y - v.y);
    }
    __syncthreads();
  }

  // C. unpack the real FFT and take |X[k]|^2 for k = 0 .. nc.
  for (int idx = tid; idx < fpb * (nc + 1); idx += 256) {
    const int fr = idx / (nc + 1), k = idx - fr * (nc + 1);
    const float2* zf = z + fr * nc;
    const float2 zk = zf[k & (nc - 1)];
    const float2 zm = zf[(nc - k) & (nc - 1)];
    const float er = 0.5f * (zk.x + zm.x), ei = 0.5f * (zk.y - zm.y);      // E = (Z[k] + conj Z[nc-k]) / 2
    const float orr = 0.5f * (zk.y + zm.y), oi = -0.5f * (zk.x - zm.x);    // O = (Z[k] - conj Z[nc-k]) / (2i)
    float2 wk;
    if (k < nc) wk = tw[k]; else wk = make_float2(-1.f, 0.f);
    const float xr = er + orr * wk.x - oi * wk.y;
    const float xi = ei + orr * wk.y + oi * wk.x;
    pw[fr * (nc + 1) + k] = xr * xr + xi * xi;
  }
  __syncthreads();

  // D. mel projection + log10; consecutive lanes = consecutive frames of one filter.
  float lmax = -3.0e38f, lmin = 3.0e38f;
  const int n_mels = a.d.n_mels;
  for (int idx = tid; idx < n_mels * fpb; idx += 256) {
    const int m = idx / fpb, fr = idx - m * fpb;
    const int f = f0 + fr;
    if (f >= a.n_frames) continue;
    const int k0 = a.d.mel_start[m], cnt = a.d.mel_count[m];
    const float* wt = a.d.mel_weight + a.d.mel_offset[m];
    const float* p = pw + fr * (nc + 1) + k0;
    float acc = 0.f;
    for (int k = 0; k < cnt; ++k) acc = fmaf(wt[k], p[k], acc);
    // mel floor: max(1e-10, x) then log10; log10(1e-10) is exactly -10 in the reference (float64 -> float32)
    const float v = acc > 1e-10f ? log10f(acc) : -10.0f;
    lmax = fmaxf(lmax, v);
    if (f < a.d.n_cols) {
      a.out[((size_t)w * n_mels + m) * a.d.n_cols + f] = v;
      lmin = fminf(lmin, v);
    }
  }
  lmax = wave_max(lmax);
  lmin = -wave_max(-lmin);
  if ((tid & 63) == 0) {
    if (lmax > -1.0e38f) atomicMax(&a.stats[2 * w + 0], f2ord(lmax));
    if (lmin < 1.0e38f) atomicMin(&a.stats[2 * w + 1], f2ord(lmin));
  }
}

// ------------------------------------------------------------------------------------------------------------------
// n_fft = 512 / 1024: wave-per-frame radix-4 FFT
// ------------------------------------------------------------------------------------------------------------------
typedef float cf2 __attribute__((ext_vector_type(2)));       // complex (re, im): adds / scalings compile to v_pk_*_f32
__device__ __forceinline__ cf2 cmul(cf2 a, cf2 w) { return (cf2){a.x, a.x} * w + (cf2){-a.y, a.y} * (cf2){w.y, w.x}; }
__device__ __forceinline__ cf2 mul_mi(cf2 a) { return (cf2){a.y, -a.x}; }      // a * (-i)
// radix-4 DIF butterfly with W4 = -i:  y_j = sum_i a_i W4^(i j)
__device__ __forceinline__ void bfly4(cf2 a[4]) {
  const cf2 s02 = a[0] + a[2], d02 = a[0] - a[2], s13 = a[1] + a[3], d13 = mul_mi(a[1] - a[3]);
  a[0] = s02 + s13; a[2] = s02 - s13; a[1] = d02 + d13; a[3] = d02 - d13;
}
// Swizzle of the 256-entry wave buffer (8-byte entries): every transpose below reads and writes 32 distinct bank pairs per
// half wave (searched exhaustively over XOR swizzles of the base-4 digits; only the mirrored read of the unpack is 2-way).
__device__ __forceinline__ int fsw(int a) { return a ^ ((((a >> 4) & 3) * 2) & 31) ^ ((((a >> 6) & 3) * 9) & 31) ^ ((((a >> 2) & 3) * 8) & 31); }
__device__ __forceinline__ void wave_lds_sync() {      // this wave's LDS writes before its own later reads (other lanes' data)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void logmel_items_kernel(wseg_logmel_desc d, LmTables* t) {
  __shared__ int cb[97];
  const int n_mels = d.n_mels;
  if (threadIdx.x == 0) {
    int n = 0;
    for (int m = 0; m < n_mels; ++m) { cb[m] = n; n += (d.mel_count[m] + LM_CH - 1) / LM_CH; }
    cb[n_mels] = n;
    t->n_items = n;
  }
  __syncthreads();
  for (int m = threadIdx.x; m <= n_mels; m += 256) t->cb[m] = cb[m];
  for (int m = threadIdx.x; m < n_mels; m += 256) {
    const int k0 = d.mel_start[m], cnt = d.mel_count[m], off = d.mel_offset[m];
    for (int c = 0; c < cnt; c += LM_CH) {
      const int n = cb[m] + c / LM_CH;
      if (n >= LM_MAX_ITEMS) break;
      t->k0[n] = k0 + c;
      for (int u = 0; u < LM_CH; ++u) t->w[n][u] = c + u < cnt ? d.mel_weight[off + c + u] : 0.f;
    }
  }
}

// NC = n_fft / 2 packed complex points per frame: 256 (n_fft 512) or 512 (n_fft 1024: one radix-2 DIF step in registers,
// z[n] +- z[n + 256], turns the frame into two 256-point FFTs — the even and the odd bins — run through the same core).
// Measurement builds only (python -m whisperseg_amd.build --stamps 5, tools/logmel_stamps.py): shader-cycle stamps of one wave over one frame
#if defined(WSEG_STAMPS) && WSEG_STAMPS == 5
__device__ unsigned long long g_lm_stamps[16];
#define WSEG_LM_STAMP(I) do { if (blockIdx.x == 3 && blockIdx.y == 1 && tid == 0 && fi == 2) g_lm_stamps[I] = __builtin_readcyclecounter(); } while (0)
#else
#define WSEG_LM_STAMP(I) do { } while (0)
#endif

template <int NC>
__global__ __launch_bounds__(256) void logmel_fft_kernel(LogmelArgs a) {
  constexpr int R = NC / 256;                                      // 256-point FFTs per frame
  constexpr int SC = 2 * R;                                        // table index of W_256^1 (table: e^{-2 pi i k / n_fft}, k < NC)
  __shared__ __attribute__((aligned(16))) cf2 s_z[4][NC];          // wave-private FFT buffers (R of 256 entries)
  __shared__ float s_pw[4][NC + 4 + LM_CH];                        // wave-private power spectra (NC + 1 bins + zero pad for whole items)
  __shared__ float s_part[4][LM_MAX_ITEMS];                        // wave-private partial mel sums
  __shared__ float s_tile[96][LM_FB + 1];                          // [filter][frame of the workgroup]
  __shared__ __attribute__((aligned(16))) float s_w[LM_MAX_ITEMS][LM_CH];     // mel weights per item, zero padded
  __shared__ int s_ik0[LM_MAX_ITEMS], s_cb[97];
  __shared__ int s_nitems;
  __shared__ cf2 s_tw[NC];                                         // the twiddle table: unpack twiddles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int w = blockIdx.y, f0 = blockIdx.x * LM_FB;
  const int n_mels = a.d.n_mels, hop = a.d.hop;
  const int64_t wstart = a.win_start[w], L = a.win_len;

  // ---- per-workgroup copy of the mel work-item tables ----
  if (tid == 0) s_nitems = a.tables->n_items;
  for (int i = tid; i <= n_mels; i += 256) s_cb[i] = a.tables->cb[i];
  for (int i = tid; i < LM_MAX_ITEMS; i += 256) s_ik0[i] = a.tables->k0[i];
  for (int i = tid; i < LM_MAX_ITEMS * LM_CH / 4; i += 256) ((float4*)&s_w[0][0])[i] = ((const float4*)&a.tables->w[0][0])[i];
  __syncthreads();
  const int n_items = s_nitems;
  for (int i = tid; i < NC; i += 256) s_tw[i] = ((const cf2*)a.d.twiddle)[i];
  for (int i = tid; i < 4 * (4 + LM_CH); i += 256) s_pw[i / (4 + LM_CH)][NC + i % (4 + LM_CH)] = 0.f;      // pad stays zero
  // ---- per-lane constants: window, twiddles ----
  const cf2* twt = (const cf2*)a.d.twiddle;                    // e^{-2 pi i k / n_fft}, k < NC; k in [NC, 2 NC): the negative
  auto twn = [&](int k) -> cf2 { const cf2 t = twt[k & (NC - 1)]; return (k & NC) ? -t : t; };
  cf2 win[4 * R], w1[3], w2[3], w3[3], wpre[R == 2 ? 4 : 1];
  const int r16 = lane & 15, r4 = lane & 3;
#pragma unroll
  for (int i = 0; i < 4 * R; ++i) {
    const int n = lane + 64 * i;
    win[i] = (cf2){a.d.window[2 * n], a.d.window[2 * n + 1]};
  }
#pragma unroll
  for (int j = 1; j < 4; ++j) { w1[j - 1] = twn(SC * j * lane); w2[j - 1] = twn(4 * SC * j * r16); w3[j - 1] = twn(16 * SC * j * r4); }
  if constexpr (R == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wpre[i] = twn(2 * (lane + 64 * i));      // W_512^n of the radix-2 step
  }
  __syncthreads();

  float* pw = s_pw[wave];
  // 256-point FFT of v (v[i] = point lane + 64 i) into zb in natural order
  auto fft256 = [&](cf2 v[4], cf2* zb) {
    bfly4(v);                                                  // stage 1 (stride 64, in registers)
#pragma unroll
    for (int j = 1; j < 4; ++j) v[j] = cmul(v[j], w1[j - 1]);
#pragma unroll
    for (int j = 0; j < 4; ++j) zb[fsw(j * 64 + lane)] = v[j];
    wave_lds_sync();
    const int g1 = (lane >> 4) * 64;                           // stage 2 (stride 16): lane = (j' = lane >> 4, r = lane & 15)
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = zb[fsw(g1 + r16 + 16 * q)];
    bfly4(v);
#pragma unroll
    for (int j = 1; j < 4; ++j) v[j] = cmul(v[j], w2[j - 1]);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) zb[fsw(g1 + j * 16 + r16)] = v[j];
    wave_lds_sync();
    const int g2 = g1 + ((lane >> 2) & 3) * 16;                // stage 3 (stride 4): lane = (j', j2 = (lane >> 2) & 3, r3 = lane & 3)
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = zb[fsw(g2 + r4 + 4 * q)];
    bfly4(v);
#pragma unroll
    for (int j = 1; j < 4; ++j) v[j] = cmul(v[j], w3[j - 1]);
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < 4; ++j) zb[fsw(g2 + j * 4 + r4)] = v[j];
    wave_lds_sync();
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = zb[fsw(g2 + r4 * 4 + q)];      // stage 4: lane = (j', j2, j3' = lane & 3): one 4-point FFT
    bfly4(v);
    wave_lds_sync();
    const int k0 = (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * r4;     // natural order: k = j' + 4 j2 + 16 j3' + 64 j4
#pragma unroll
    for (int j = 0; j < 4; ++j) zb[fsw(k0 + 64 * j)] = v[j];
  };
  // bin k of the NC-point FFT: R == 1: zb[k]; R == 2: even bins in the first, odd bins in the second 256-entry buffer
  auto zget = [&](int k) -> cf2 {
    if constexpr (R == 1) return s_z[wave][fsw(k)];
    else return s_z[wave][(k & 1) * 256 + fsw(k >> 1)];
  };
  float lmax = -3.0e38f, lmin = 3.0e38f;
  for (int fi = 0; fi < LM_FB / 4; ++fi) {
    const int fr = wave * (LM_FB / 4) + fi, f = f0 + fr;       // wave-uniform
    if (f >= a.n_frames) break;
    WSEG_LM_STAMP(0);
    // ---- A. load + window: z[n] = x[2n] + i x[2n+1], n = lane + 64 i ----
    cf2 v[4 * R];
    const int64_t base = (int64_t)f * hop - NC;                // window-local index of sample 0 of the frame
    const int64_t abase = wstart + base;
    if (base >= 0 && base + 2 * NC <= L && abase >= 0 && abase + 2 * NC <= a.n_audio) {      // wave-uniform: the frame is interior
      typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
      const float* src = a.audio + abase + 2 * lane;
#pragma unroll
      for (int i = 0; i < 4 * R; ++i) { const f2u t = *(const f2u*)(src + 128 * i); v[i] = (cf2){t.x, t.y} * win[i]; }
    } else
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) {
      float xs[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        int64_t jl = base + 2 * (lane + 64 * i) + q;
        if (jl < 0) jl = -jl;                                  // reflect (np.pad mode="reflect")
        if (jl >= L) jl = 2 * (L - 1) - jl;
        const int64_t ai = wstart + jl;
        xs[q] = (ai >= 0 && ai < a.n_audio) ? a.audio[ai] : 0.f;
      }
      v[i] = (cf2){xs[0], xs[1]} * win[i];
    }
    WSEG_LM_STAMP(1);
    // ---- B. FFT ----
    if constexpr (R == 1) {
      fft256(v, s_z[wave]);
    } else {
      cf2 ev[4], od[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { ev[i] = v[i] + v[i + 4]; od[i] = cmul(v[i] - v[i + 4], wpre[i]); }
      fft256(ev, s_z[wave]);
      wave_lds_sync();
      fft256(od, s_z[wave] + 256);
    }
    wave_lds_sync();
    WSEG_LM_STAMP(2);
    // ---- C. unpack the real FFT, |X[k]|^2 for k = lane + 64 i (and k = NC in lane 0) ----
#pragma unroll
    for (int i = 0; i < 4 * R; ++i) {
      const int k = lane + 64 * i;
      const cf2 zk = zget(k), zm = zget((NC - k) & (NC - 1));
      const cf2 e = (cf2){0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};        // E = (Z[k] + conj Z[nc-k]) / 2
      const cf2 o = (cf2){0.5f * (zk.y + zm.y), -0.5f * (zk.x - zm.x)};       // O = (Z[k] - conj Z[nc-k]) / (2i)
      const cf2 x = e + cmul(o, s_tw[k]);
      pw[k] = x.x * x.x + x.y * x.y;
      if (k == 0) { const float xn = zk.x - zk.y; pw[NC] = xn * xn; }         // X[NC] = Re Z[0] - Im Z[0]
    }
    wave_lds_sync();
    WSEG_LM_STAMP(3);
    // ---- D. mel projection: items of <= LM_CH bins, then each filter sums its items in order ----
    for (int it = lane; it < n_items; it += 64) {
      const int kk = s_ik0[it];
      const float4 wa = *(const float4*)&s_w[it][0], wb = *(const float4*)&s_w[it][4];
      const float* pp = pw + kk;
      float acc = wa.x * pp[0];
      acc = fmaf(wa.y, pp[1], acc); acc = fmaf(wa.z, pp[2], acc); acc = fmaf(wa.w, pp[3], acc);
      acc = fmaf(wb.x, pp[4], acc); acc = fmaf(wb.y, pp[5], acc); acc = fmaf(wb.z, pp[6], acc); acc = fmaf(wb.w, pp[7], acc);
      s_part[wave][it] = acc;
    }
    wave_lds_sync();
    WSEG_LM_STAMP(4);
    for (int m = lane; m < n_mels; m += 64) {
      float acc = 0.f;
      for (int c = s_cb[m]; c < s_cb[m + 1]; ++c) acc += s_part[wave][c];
      // mel floor: max(1e-10, x) then log10; log10(1e-10) is exactly -10 in the reference (float64 -> float32)
      const float val = acc > 1e-10f ? __log2f(acc) * 0.30102999566398120f : -10.0f;      // v_log_f32: ~1 ulp of log2
      s_tile[m][fr] = val;
      lmax = fmaxf(lmax, val);
      if (f < a.d.n_cols) lmin = fminf(lmin, val);
    }
    wave_lds_sync();
    WSEG_LM_STAMP(5);
  }
  lmax = wave_max(lmax);
  lmin = -wave_max(-lmin);
  if (lane == 0) {
    if (lmax > -1.0e38f) atomicMax(&a.stats[2 * w + 0], f2ord(lmax));
    if (lmin < 1.0e38f) atomicMin(&a.stats[2 * w + 1], f2ord(lmin));
  }
  __syncthreads();
  // ---- E. 32 consecutive frames of every filter = one 128-byte row segment of the output image ----
  const int n_valid = min(min(a.n_frames, a.d.n_cols) - f0, LM_FB);
  for (int idx = tid; idx < n_mels * LM_FB; idx += 256) {
    const int m = idx / LM_FB, fr = idx - m * LM_FB;
    if (fr < n_valid) a.out[((size_t)w * n_mels + m) * a.d.n_cols + f0 + fr] = s_tile[m][fr];
  }
}

// out[w][m][c] = (max(out, wmax - 8) + 4) / 4 in place; columns >= n_frames take the window minimum
// (reference model.py:155-161).
__global__ __launch_bounds__(256) void logmel_finish_kernel(const uint32_t* __restrict__ stats, float* __restrict__ out, int n_mels,
                                                            int n_frames, int n_cols) {
  const int w = blockIdx.y;
  const float wmax = ord2f(stats[2 * w + 0]);
  const uint32_t mn = stats[2 * w + 1];
  const float floorv = wmax - 8.0f;
  const float fill = (n_frames > 0 && mn != 0xffffffffu) ? (fmaxf(ord2f(mn), floorv) + 4.0f) / 4.0f : 0.f;
  const int total = n_mels * n_cols;
  float* o = out + (size_t)w * total;
  if ((n_cols & 3) == 0) {                      // 16-byte accesses: a 4-column group never straddles a row
    for (int idx = (blockIdx.x * 256 + threadIdx.x) * 4; idx < total; idx += gridDim.x * 1024) {
      const int c = idx % n_cols;
      float4 v = *(const float4*)(o + idx);
      v.x = c + 0 < n_frames ? (fmaxf(v.x, floorv) + 4.0f) / 4.0f : fill;
      v.y = c + 1 < n_frames ? (fmaxf(v.y, floorv) + 4.0f) / 4.0f : fill;
      v.z = c + 2 < n_frames ? (fmaxf(v.z, floorv) + 4.0f) / 4.0f : fill;
      v.w = c + 3 < n_frames ? (fmaxf(v.w, floorv) + 4.0f) / 4.0f : fill;
      *(float4*)(o + idx) = v;
    }
  } else {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
      const int c = idx % n_cols;
      o[idx] = c < n_frames ? (fmaxf(o[idx], floorv) + 4.0f) / 4.0f : fill;
    }
  }
}

static int frames_per_block(int n_fft) {
  int fpb = 16 * 512 / n_fft;   // 32 KiB of packed complex frames per workgroup
  return fpb < 1 ? 1 : fpb;
}

}  // namespace wseg

using namespace wseg;

extern "C" size_t wseg_logmel_scratch_bytes(const wseg_logmel_desc* d, int32_t n_windows, int64_t win_len) {
  if (!d || d->hop <= 0 || n_windows <= 0) return 0;
  (void)win_len;
  // two ordered-uint words per window (max / min) + the mel work-item tables of the fast path; no raw spectrum any more
  return align_up((size_t)n_windows * 8, 256) + align_up(sizeof(LmTables), 256);
}

extern "C" int wseg_logmel_f32(const wseg_logmel_desc* d, const float* audio, int64_t n_audio,
                               const int64_t* win_start, int32_t n_windows, int64_t win_len,
                               void* scratch, size_t scratch_bytes, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d || !win_start || !out || !scratch) { set_error("wseg_logmel_f32: null argument"); return WSEG_ERR_INVALID; }
  if (n_windows <= 0) return WSEG_OK;
  int lg = 0;
  while ((1 << lg) < d->n_fft) ++lg;
  if ((1 << lg) != d->n_fft || d->n_fft < 64 || d->n_fft > 8192) { set_error("n_fft %d unsupported", d->n_fft); return WSEG_ERR_INVALID; }
  if (d->hop <= 0 || win_len <= d->n_fft / 2) { set_error("hop %d / win_len %lld unsupported", d->hop, (long long)win_len); return WSEG_ERR_INVALID; }
  if (scratch_bytes < wseg_logmel_scratch_bytes(d, n_windows, win_len)) { set_error("logmel scratch too small"); return WSEG_ERR_STATE; }
  const int64_t nf64 = win_len / d->hop;
  if (nf64 > (1 << 24)) { set_error("too many frames per window"); return WSEG_ERR_INVALID; }
  LogmelArgs a;
  a.d = *d;
  a.audio = audio;
  a.n_audio = audio ? n_audio : 0;
  a.win_start = win_start;
  a.win_len = win_len;
  a.n_frames = (int32_t)nf64;
  a.fpb = frames_per_block(d->n_fft);
  a.lg_nc = lg - 1;
  a.out = out;
  a.stats = (uint32_t*)scratch;
  a.tables = (const LmTables*)((char*)scratch + align_up((size_t)n_windows * 8, 256));
  // stats: max = lowest ordered value (0), min = highest (0xffffffff)
  WSEG_HIP_CHECK(hipMemsetAsync(a.stats, 0, (size_t)n_windows * 8, stream));
  {
    // set the min slots to 0xffffffff with a strided 2D memset
    WSEG_HIP_CHECK(hipMemset2DAsync((char*)a.stats + 4, 8, 0xff, 4, (size_t)n_windows, stream));
  }
  const int nc = d->n_fft / 2;
  const size_t smem = (size_t)nc * 8 + (size_t)a.fpb * nc * 8 + (size_t)a.fpb * (nc + 1) * 4;
  static const bool generic_only = getenv("WSEG_LOGMEL_GENERIC") != nullptr;       // A/B + test knob
  bool fast_ok = (d->n_fft == 512 || d->n_fft == 1024) && d->n_mels <= 96 && !generic_only;
  if (a.n_frames > 0 && fast_ok) {
    dim3 grid(cdiv(a.n_frames, LM_FB), n_windows);
    hipLaunchKernelGGL(logmel_items_kernel, dim3(1), dim3(256), 0, stream, *d, (LmTables*)a.tables);
    if (d->n_fft == 512) hipLaunchKernelGGL(logmel_fft_kernel<256>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(logmel_fft_kernel<512>, grid, dim3(256), 0, stream, a);
    WSEG_LAUNCH_CHECK();
  } else if (a.n_frames > 0) {
    dim3 grid(cdiv(a.n_frames, a.fpb), n_windows);
    // the function attribute is per device (thread-per-device mode, reference model.py:173-184): set once for each
    static bool attr_set[64] = {};
    int dev = 0;
    WSEG_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
      WSEG_HIP_CHECK(hipFuncSetAttribute((const void*)logmel_stft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(logmel_stft_kernel, grid, dim3(256), smem, stream, a);
    WSEG_LAUNCH_CHECK();
  }
  {
    const int total = d->n_mels * d->n_cols;
    dim3 grid(cdiv(total, 256 * 4), n_windows);
    hipLaunchKernelGGL(logmel_finish_kernel, grid, dim3(256), 0, stream, a.stats, out, d->n_mels, a.n_frames, d->n_cols);
    WSEG_LAUNCH_CHECK();
  }
  return WSEG_OK;
}

#if defined(WSEG_STAMPS) && WSEG_STAMPS == 5
extern "C" int wseg_debug_logmel_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(wseg::g_lm_stamps), sizeof(unsigned long long) * 16);
}
#endif
