// Polyphase FIR resampler (rational up/down): y[m] = sum_k h[(m + pre_remove) * down - pre_pad - k * up] * x[k].
// the reference resamples with librosa.load(path, sr=target) before segment()
// (scripts/segment.py:48,61; evaluate.py:58) — an un-pinned third-party resampler.  This kernel implements the standard
// Kaiser-windowed-sinc polyphase structure (filter designed on the host, see whisperseg_amd/resample.py).
// HBM-bound: 4 B in per input sample + 4 B out per output sample; the taps (<= 35 KiB) stay in L1/L2.
#include "wseg_common.h"

namespace wseg {

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, long long n_in, const float* __restrict__ h, int n_h,
                                                       int up, int down, int pre_pad, int pre_remove, float* __restrict__ y,
                                                       long long n_out) {
  for (long long m = (long long)blockIdx.x * 256 + threadIdx.x; m < n_out; m += (long long)gridDim.x * 256) {
    const long long c = (m + pre_remove) * (long long)down - pre_pad;
    long long k_hi = c / up;                       // floor (c >= -pre_pad; handle negatives below)
    if (c < 0) k_hi = -((-c + up - 1) / up);
    long long lo_num = c - n_h + 1;
    long long k_lo = lo_num <= 0 ? 0 : (lo_num + up - 1) / up;
    if (k_hi > n_in - 1) k_hi = n_in - 1;
    float acc = 0.f;
    for (long long k = k_lo; k <= k_hi; ++k) acc = fmaf(h[c - k * up], x[k], acc);
    y[m] = acc;
  }
}

}  // namespace wseg

using namespace wseg;

extern "C" int wseg_resample_f32(const float* x, int64_t n_in, const float* taps, int32_t n_taps, int32_t up, int32_t down,
                                 int32_t pre_pad, int32_t pre_remove, float* y, int64_t n_out, void* stream_) {
  hipStream_t s = (hipStream_t)stream_;
  if (!x || !taps || !y || n_in < 0 || n_out < 0 || up <= 0 || down <= 0 || n_taps <= 0) { set_error("wseg_resample_f32: bad argument"); return WSEG_ERR_INVALID; }
  if (n_out == 0) return WSEG_OK;
  long long blocks = (n_out + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, (long long)n_in, taps, n_taps, up, down, pre_pad,
                     pre_remove, y, (long long)n_out);
  WSEG_LAUNCH_CHECK();
  return WSEG_OK;
}
