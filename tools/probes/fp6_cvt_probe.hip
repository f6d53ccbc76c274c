// Probe of the gfx950 fp6 (e2m3) conversion instructions the mixed split-precision mode relies on:
//   v_cvt_scalef32_pk32_fp6_f16 / v_cvt_scalef32_2xpk16_fp6_f32 (quantise 32 values of ONE lane with a float scale),
//   v_cvt_scalef32_pk32_f32_fp6 (decode), and their agreement with v_mfma_scale_f32_16x16x128_f8f6f4 (cbsz = blgp = 2) + e8m0 scales.
// Questions answered: is the stored value x / scale or x * scale?  rounding (nearest even?) and saturation?  bit order of the
// 32 x 6-bit codes?  does MFMA(fp6(A), scale byte) . ones == sum of decode(A)?
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v32f __attribute__((ext_vector_type(32)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void cvt_kernel(const float* in, float scale, unsigned* raw_h, unsigned* raw_f, float* back) {
  const int l = threadIdx.x;
  v32h h; v16f a, b;
  for (int i = 0; i < 32; ++i) h[i] = (_Float16)in[l * 32 + i];
  for (int i = 0; i < 16; ++i) { a[i] = in[l * 32 + i]; b[i] = in[l * 32 + 16 + i]; }
  const v6u r1 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, scale);
  const v6u r2 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
  for (int i = 0; i < 6; ++i) { raw_h[l * 6 + i] = r1[i]; raw_f[l * 6 + i] = r2[i]; }
  const v32f d = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(r1, scale);
  for (int i = 0; i < 32; ++i) back[l * 32 + i] = d[i];
}

// D[i][j] = sum_k A[i][k] B[j][k]: A rows quantised per lane (row l & 15, chunk l >> 4) with per-lane scale 2^e_l, B = all ones (code of 1.0)
__global__ void mfma_kernel(const float* in, const int* expo, float* D, float* lane_sum) {
  const int l = threadIdx.x;
  v32h h;
  for (int i = 0; i < 32; ++i) h[i] = (_Float16)in[((l & 15) * 4 + (l >> 4)) * 32 + i];
  const int e = expo[l];
  const v6u r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, ldexpf(1.0f, e));
  {
    const v32f d = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(r, ldexpf(1.0f, e));
    float sum = 0.f;
    for (int i = 0; i < 32; ++i) sum += d[i];
    lane_sum[l] = sum;      // what this lane's 32 quantised elements are worth
  }
  v32h ones;
  for (int i = 0; i < 32; ++i) ones[i] = (_Float16)1.0f;
  const v6u o = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(ones, 1.0f);
  const v8i a = {(int)r[0], (int)r[1], (int)r[2], (int)r[3], (int)r[4], (int)r[5], 0, 0};
  const v8i b = {(int)o[0], (int)o[1], (int)o[2], (int)o[3], (int)o[4], (int)o[5], 0, 0};
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, 127 + e, 0, 127);
  for (int q = 0; q < 4; ++q) D[((l >> 4) * 4 + q) * 16 + (l & 15)] = c[q];
}

static float e2m3(unsigned c) {
  const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, e - 1);
  return s ? -v : v;
}

int main() {
  const float vals[32] = {0.f, 0.0625f, 0.125f, 0.1875f, 0.3f, 0.5f, 0.9375f, 1.0f, 1.0625f, 1.1875f, 1.5f, 1.9f, 2.0f, 2.125f, 2.375f, 3.0f,
                          3.9f, 4.0f, 4.25f, 4.75f, 6.0f, 7.0f, 7.5f, 7.9f, 9.0f, 30.0f, -1.0f, -2.125f, -7.5f, -100.0f, 0.03f, -0.0624f};
  std::vector<float> in(64 * 32);
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 32; ++i) in[l * 32 + i] = vals[i];
  float* din; unsigned *dh, *df; float* dback;
  hipMalloc(&din, in.size() * 4); hipMalloc(&dh, 64 * 6 * 4); hipMalloc(&df, 64 * 6 * 4); hipMalloc(&dback, 64 * 32 * 4);
  hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  for (float scale : {1.0f, 4.0f, 0.25f}) {
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, din, scale, dh, df, dback);
    unsigned rh[6], rf[6]; float back[32];
    hipMemcpy(rh, dh, 24, hipMemcpyDeviceToHost); hipMemcpy(rf, df, 24, hipMemcpyDeviceToHost); hipMemcpy(back, dback, 128, hipMemcpyDeviceToHost);
    printf("scale %.2f   (f16 path raw == f32 path raw: %s)\n", scale, memcmp(rh, rf, 24) == 0 ? "yes" : "NO");
    for (int i = 0; i < 32; ++i) {
      const int bit = 6 * i;
      unsigned long long two = rh[bit >> 5] | ((unsigned long long)(bit >> 5 < 5 ? rh[(bit >> 5) + 1] : 0) << 32);
      const unsigned code = (unsigned)(two >> (bit & 31)) & 63;
      printf("  x %9.4f -> code %2u (little-endian e2m3 %7.4f)  decode %9.4f   x/scale %9.4f\n", vals[i], code, e2m3(code), back[i], vals[i] / scale);
    }
  }
  // MFMA agreement: random values, per-lane exponents
  std::vector<float> rnd(64 * 32); std::vector<int> ex(64);
  srand(3);
  for (int l = 0; l < 64; ++l) { ex[l] = (l % 7) - 3; for (int i = 0; i < 32; ++i) rnd[l * 32 + i] = ldexpf((rand() % 121 - 60) / 8.0f, ex[l]); }
  int* dex; float* dD;
  hipMalloc(&dex, 256); hipMalloc(&dD, 1024);
  hipMemcpy(din, rnd.data(), rnd.size() * 4, hipMemcpyHostToDevice);
  // lane l of the MFMA kernel reads input row ((l & 15) * 4 + (l >> 4)): give that row the exponent of lane l
  std::vector<int> exl(64);
  for (int l = 0; l < 64; ++l) exl[l] = ex[(l & 15) * 4 + (l >> 4)];
  hipMemcpy(dex, exl.data(), 256, hipMemcpyHostToDevice);
  float* dls;
  hipMalloc(&dls, 256);
  hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, din, dex, dD, dls);
  std::vector<float> D(256), ls(64);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(ls.data(), dls, 256, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int i = 0; i < 16; ++i) {
    double ref = 0;      // the decoded (quantised) values of the four lanes (i, g) that hold row i
    for (int g = 0; g < 4; ++g) ref += ls[g * 16 + i];
    for (int j = 0; j < 16; ++j) maxerr = fmax(maxerr, fabs(D[i * 16 + j] - ref));
    if (i < 3) printf("row %d: MFMA %.6f  sum of decoded %.6f\n", i, D[i * 16], ref);
  }
  printf("MFMA(fp6(A) with per-lane e8m0 scales) . ones: max |err| = %.3g -> %s\n", maxerr, maxerr < 1e-3 ? "OK" : "MISMATCH");
  return 0;
}
