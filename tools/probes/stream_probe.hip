// HBM read-stream probe: how fast can 256-thread workgroups pull contiguous 64-KiB slabs, (a) with 16-byte loads into
// VGPRs, (b) with LDS-DMA (global_load_lds) into a 64-KiB LDS buffer.   hipcc --offload-arch=gfx950 -O3 stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_vgpr(const uint4* __restrict__ src, float* __restrict__ out, int slabs_per_wg) {
  float acc = 0.f;
  for (int s = 0; s < slabs_per_wg; ++s) {
    const uint4* p = src + ((size_t)blockIdx.x * slabs_per_wg + s) * 4096;     // 64 KiB = 4096 x 16 B
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      typedef unsigned u4 __attribute__((ext_vector_type(4)));
      const u4 v = __builtin_nontemporal_load((const u4*)(p + i * 256 + threadIdx.x));
      acc += __uint_as_float(v[0] ^ v[1] ^ v[2] ^ v[3]);
    }
  }
  if (acc == 1234.5f) out[0] = acc;
}

template <int DEPTH>   // DEPTH 16-KiB chunks of the 64-KiB buffer in flight
__global__ __launch_bounds__(256) void k_dma(const uint4* __restrict__ src, float* __restrict__ out, int slabs_per_wg) {
  __shared__ __attribute__((aligned(16))) uint4 buf[4096];
  const int wave = threadIdx.x >> 6;
  float acc = 0.f;
  const int chunks = slabs_per_wg * 4;                     // 16-KiB chunks: 4 glds per thread each
  auto issue = [&](int c) {
    const uint4* p = src + ((size_t)blockIdx.x * slabs_per_wg * 4 + c) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i * 256 + threadIdx.x),
                                       (__attribute__((address_space(3))) void*)(buf + (c & 3) * 1024 + i * 256 + wave * 64), 16, 0, 0);
  };
  for (int c = 0; c < DEPTH && c < chunks; ++c) issue(c);
  for (int c = 0; c < chunks; ++c) {
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DEPTH == 2) { if (c + 1 < chunks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else { if (c + 2 < chunks) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (c + 1 < chunks) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    // touch one value of the landed chunk (own lane's piece) so the stream is consumed
    unsigned t;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((unsigned)(((c & 3) * 1024 + wave * 64 + (threadIdx.x & 63)) * 16)) : "memory");
    acc += __uint_as_float(t);
    if (c + DEPTH < chunks) issue(c + DEPTH);              // same wave re-fills only the pieces it read itself
  }
  if (acc == 1234.5f) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t total = (size_t)2 << 30;                    // 2 GiB
  uint4* src; float* out;
  CHECK(hipMalloc(&src, total)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(src, 1, total));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int spw : {1, 2, 4}) {
    const int grid = (int)(total / 65536 / spw);
    auto run = [&](const char* name, auto launch) {
      launch(); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < 5; ++i) launch();
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("%-28s slabs/wg %d grid %6d: %.2f TB/s\n", name, spw, grid, total * 5 / (ms * 1e-3) / 1e12);
    };
    run("vgpr 16-B nt loads", [&] { hipLaunchKernelGGL(k_vgpr, dim3(grid), dim3(256), 0, 0, src, out, spw); });
    run("lds-dma, 1 chunk in flight", [&] { hipLaunchKernelGGL(k_dma<1>, dim3(grid), dim3(256), 0, 0, src, out, spw); });
    run("lds-dma, 2 chunks in flight", [&] { hipLaunchKernelGGL(k_dma<2>, dim3(grid), dim3(256), 0, 0, src, out, spw); });
    run("lds-dma, 3 chunks in flight", [&] { hipLaunchKernelGGL(k_dma<3>, dim3(grid), dim3(256), 0, 0, src, out, spw); });
  }
  return 0;
}
