// Log-mel front-end for gfx950: framed STFT (periodic Hann, reflect-centred) -> |X|^2 -> slaney mel
// -> log10 -> per-window (max - 8) clamp -> (x + 4) / 4, for every window of a recording at once.
//
// Math follows HF feature_extraction_whisper.py:105-133 / audio_utils.py:809-1017 as configured by
// reference audio_utils.py:45-76; window slicing, zero padding, truncation to total_spec_columns and
// the min-fill follow reference model.py:138-161.
//
// Bound: HBM (algorithmic bytes per window = 4*win_len in + 4*n_mels*n_cols out); the FFT itself is
// ~0.05 GFLOP/window.  Layout: one workgroup = FPB consecutive frames of one window; samples are read
// coalesced (consecutive lanes -> consecutive samples), the packed real FFT (n_fft/2 complex points,
// radix-2 in place) runs in LDS with the twiddle table staged in LDS once per workgroup, and the mel
// projection walks each filter's contiguous non-zero bin range (sparse triangles, <= 2 filters per bin).
#include "wseg_common.h"

namespace wseg {

__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

struct LogmelArgs {
  wseg_logmel_desc d;
  const float* audio;
  int64_t n_audio;
  const int64_t* win_start;
  int64_t win_len;
  int32_t n_frames;   // floor(win_len / hop): frames kept after HF drops the last one
  int32_t fpb;        // frames per workgroup
  int32_t lg_nc;      // log2(n_fft / 2)
  float* raw;         // [W][n_mels][n_frames] log10 mel
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
    a.raw[((size_t)w * n_mels + m) * a.n_frames + f] = v;
    lmax = fmaxf(lmax, v);
    if (f < a.d.n_cols) lmin = fminf(lmin, v);
  }
  lmax = wave_max(lmax);
  lmin = -wave_max(-lmin);
  if ((tid & 63) == 0) {
    if (lmax > -1.0e38f) atomicMax(&a.stats[2 * w + 0], f2ord(lmax));
    if (lmin < 1.0e38f) atomicMin(&a.stats[2 * w + 1], f2ord(lmin));
  }
}

// out[w][m][c] = (max(raw, wmax - 8) + 4) / 4 ; columns >= n_frames take the window minimum
// (reference model.py:155-161).
__global__ __launch_bounds__(256) void logmel_finish_kernel(const float* __restrict__ raw, const uint32_t* __restrict__ stats,
                                                            float* __restrict__ out, int n_mels, int n_frames, int n_cols) {
  const int w = blockIdx.y;
  const float wmax = ord2f(stats[2 * w + 0]);
  const uint32_t mn = stats[2 * w + 1];
  const float floorv = wmax - 8.0f;
  const float fill = (n_frames > 0 && mn != 0xffffffffu) ? (fmaxf(ord2f(mn), floorv) + 4.0f) / 4.0f : 0.f;
  const int total = n_mels * n_cols;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int m = idx / n_cols, c = idx - m * n_cols;
    float v = fill;
    if (c < n_frames) v = (fmaxf(raw[((size_t)w * n_mels + m) * n_frames + c], floorv) + 4.0f) / 4.0f;
    out[(size_t)w * total + idx] = v;
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
  const int64_t nf = win_len / d->hop;
  return align_up((size_t)n_windows * d->n_mels * (size_t)nf * sizeof(float), 256) + align_up((size_t)n_windows * 8, 256);
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
  a.raw = (float*)scratch;
  a.stats = (uint32_t*)((char*)scratch + align_up((size_t)n_windows * d->n_mels * (size_t)nf64 * sizeof(float), 256));
  // stats: max = lowest ordered value (0), min = highest (0xffffffff)
  WSEG_HIP_CHECK(hipMemsetAsync(a.stats, 0, (size_t)n_windows * 8, stream));
  {
    // set the min slots to 0xffffffff with a strided 2D memset
    WSEG_HIP_CHECK(hipMemset2DAsync((char*)a.stats + 4, 8, 0xff, 4, (size_t)n_windows, stream));
  }
  const int nc = d->n_fft / 2;
  const size_t smem = (size_t)nc * 8 + (size_t)a.fpb * nc * 8 + (size_t)a.fpb * (nc + 1) * 4;
  if (a.n_frames > 0) {
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
    hipLaunchKernelGGL(logmel_finish_kernel, grid, dim3(256), 0, stream, a.raw, a.stats, out, d->n_mels, a.n_frames, d->n_cols);
    WSEG_LAUNCH_CHECK();
  }
  return WSEG_OK;
}
