// Is v_pk_fma_f32 bit-equal to v_fma_f32 lane by lane?  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off pkfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* q, const float* kk, unsigned* bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float a = 0.f, b = 0.f;
  f2 ab = {0.f, 0.f};
  for (int e = 0; e < 8; ++e) {
    const float q0 = q[(i * 8 + e) * 2], q1 = q[(i * 8 + e) * 2 + 1], kv = kk[i * 8 + e];
    a = fmaf(q0, kv, a);
    b = fmaf(q1, kv, b);
    const f2 qq = {q0, q1}, k2 = {kv, kv};
    ab = __builtin_elementwise_fma(qq, k2, ab);
  }
  if (__float_as_uint(a) != __float_as_uint(ab[0]) || __float_as_uint(b) != __float_as_uint(ab[1])) atomicAdd(bad, 1u);
}
int main() {
  const int n = 1 << 20;
  float *q, *kk; unsigned* bad;
  hipMallocManaged(&q, n * 16 * 4); hipMallocManaged(&kk, n * 8 * 4); hipMallocManaged(&bad, 4);
  srand(1);
  for (int i = 0; i < n * 16; ++i) { unsigned u = ((unsigned)rand() << 16) & 0xffff0000u; u = (u & 0x807f0000u) | 0x3f000000u; q[i] = *(float*)&u * (rand() % 2 ? 1 : -1); }
  for (int i = 0; i < n * 8; ++i) { unsigned u = ((unsigned)rand() << 16) & 0xffff0000u; u = (u & 0x807f0000u) | 0x3f800000u; kk[i] = *(float*)&u; }
  *bad = 0;
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, q, kk, bad);
  hipDeviceSynchronize();
  printf("mismatching chains: %u of %d\n", *bad, n);
  return 0;
}
