#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../whisperseg_amd/csrc/wseg_common.h"
using namespace wseg;
__global__ void k(unsigned* out) {
  unsigned v = threadIdx.x * 7 + 3;
  out[0 * 64 + threadIdx.x] = lane_xor_u<1>(v);
  out[1 * 64 + threadIdx.x] = lane_xor_u<2>(v);
  out[2 * 64 + threadIdx.x] = lane_xor_u<4>(v);
  out[3 * 64 + threadIdx.x] = lane_xor_u<8>(v);
  out[4 * 64 + threadIdx.x] = lane_xor_u<16>(v);
  out[5 * 64 + threadIdx.x] = lane_xor_u<32>(v);
}
int main() {
  unsigned* d; hipMalloc(&d, 6 * 64 * 4);
  k<<<1, 64>>>(d);
  unsigned h[6 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int m = 0; m < 6; ++m) for (int l = 0; l < 64; ++l) if (h[m * 64 + l] != (unsigned)((l ^ (1 << m)) * 7 + 3)) { ++bad; if (bad < 10) printf("M=%d lane %d got %u\n", 1 << m, l, h[m * 64 + l]); }
  printf("lane_xor check: %d mismatches\n", bad);
  return bad != 0;
}
