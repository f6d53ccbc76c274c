// Probe of the gfx950 block-scaled MX matrix-core instruction used by the mixed split-precision mode (csrc/wseg_gemm.hip):
//   v_mfma_scale_f32_16x16x128_f8f6f4 with bf8 (e5m2) operands and CONSTANT e8m0 scales.
// (1) semantics: D[i][j] = C[i][j] + 2^(sa-127) 2^(sb-127) sum_k A[i][k] B[j][k], lane l holds row (l & 15), 32 consecutive
//     bytes of chunk (l >> 4) — checked against a host reference with an asymmetric random B;
// (2) rate: a dependent-free stream of MX MFMAs against v_mfma_f32_16x16x32_f16 (cycles per instruction on one SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o mx_mfma_probe mx_mfma_probe.hip && ./mx_mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

static float e5m2_to_float(uint8_t b) {
  const int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
  float v;
  if (e == 0) v = ldexpf((float)m, -16);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = ldexpf(1.0f + m / 4.0f, e - 15);
  return s ? -v : v;
}

__global__ void probe_kernel(const uint8_t* A, const uint8_t* B, float* D, int sa, int sb) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  const v8i a = *(const v8i*)(A + r * 128 + g * 32);
  const v8i b = *(const v8i*)(B + r * 128 + g * 32);
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 1, 0, sa, 0, sb);
  // standard 16x16 C/D map: col = lane & 15, row = (lane >> 4) * 4 + reg
  for (int q = 0; q < 4; ++q) D[(g * 4 + q) * 16 + r] = c[q];
}

template <int WHICH>
__global__ void rate_kernel(unsigned long long* out, int iters) {
  const int lane = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x3c3c3c3c + lane + i; b[i] = 0x38383838 + lane * 3 + i; }
  v8h ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * (lane + i)); hb[i] = (_Float16)(0.02f * (lane - i)); }
  v4f acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma nounroll
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (WHICH == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 1, 1, 0, 116, 0, 127);
      else if constexpr (WHICH == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
      else if constexpr (WHICH == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 2, 2, 0, 116, 0, 127);   // fp6 e2m3
      else {      // the mixed mode's group: 4 f16 MFMAs + 2 MX bf8 MFMAs = 128 logical k of one 16x16 tile
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, ha, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, hb, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 1, 1, 0, 116, 0, 127);
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, acc[i], 1, 1, 0, 127, 0, 116);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  if (lane == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(s != 12345.f); }
}

int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  srand(7);
  for (auto& v : A) { v = (uint8_t)(rand() & 0xff); if (((v >> 2) & 31) == 31) v &= 0x7b; }   // no inf / nan
  for (auto& v : B) { v = (uint8_t)(rand() & 0xff); if (((v >> 2) & 31) == 31) v &= 0x7b; }
  for (auto& v : A) if (((v >> 2) & 31) > 20) v = (v & 0x83) | (18 << 2);                        // keep magnitudes moderate
  for (auto& v : B) if (((v >> 2) & 31) > 20) v = (v & 0x83) | (17 << 2);
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  const int cases[3][2] = {{127, 127}, {116, 127}, {127, 116}};
  for (auto& sc : cases) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, sc[0], sc[1]);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int k = 0; k < 128; ++k) ref += (double)e5m2_to_float(A[i * 128 + k]) * e5m2_to_float(B[j * 128 + k]);
        ref *= ldexp(1.0, sc[0] - 127 + sc[1] - 127);
        // which operand is the row?  try both conventions
        const double e1 = fabs(D[i * 16 + j] - ref), e2 = fabs(D[j * 16 + i] - ref);
        maxerr = fmax(maxerr, e1);
        maxref = fmax(maxref, fabs(ref));
        if (i == 1 && j == 2) printf("  D[1][2] %.6g  D[2][1] %.6g  ref(A row 1 . B row 2) %.6g  (err row=A %.3g, row=B %.3g)\n", D[i * 16 + j], D[j * 16 + i], ref, e1, e2);
      }
    printf("scales (%d, %d): max |D[i][j] - ref| = %.4g on max |ref| %.4g  -> %s\n", sc[0], sc[1], maxerr, maxref, maxerr <= 2e-4 * maxref ? "row = A operand: OK (fp32 accumulation noise)" : "MISMATCH (see D[2][1])");
  }
  unsigned long long* dT;
  hipMalloc(&dT, 16 * 1024);
  const char* names[4] = {"mx bf8 16x16x128", "f16 16x16x32", "mx fp6 16x16x128", "group 4 f16 + 2 mx bf8"};
  for (int which = 0; which < 4; ++which) {
    for (int waves = 1; waves <= 2; ++waves) {
      const dim3 grid(256), block(64 * 4 * waves);      // every CU busy: the clock the chip sustains under this load
      for (int rep = 0; rep < 2; ++rep) {
        if (which == 0) hipLaunchKernelGGL(rate_kernel<0>, grid, block, 0, 0, dT, 4000);
        else if (which == 1) hipLaunchKernelGGL(rate_kernel<1>, grid, block, 0, 0, dT, 4000);
        else if (which == 2) hipLaunchKernelGGL(rate_kernel<2>, grid, block, 0, 0, dT, 4000);
        else hipLaunchKernelGGL(rate_kernel<3>, grid, block, 0, 0, dT, 4000);
        hipDeviceSynchronize();
      }
      unsigned long long t[2];
      hipMemcpy(t, dT, 16, hipMemcpyDeviceToHost);
      printf("%-24s %d wave(s) per SIMD, 256 workgroups: %.1f cycles per loop body of 8 accumulators per wave (= %.1f per accumulator step)\n",
             names[which], waves, (double)t[0] / 4000.0, (double)t[0] / 4000.0 / 8);
    }
  }
  return 0;
}
