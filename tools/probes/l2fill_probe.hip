// Per-CU operand-fill probe: how fast can ONE workgroup per CU pull an L2-resident region (a) into LDS with LDS-DMA
// (global_load_lds, 16 B per lane, the GEMM kernels' staging path) at 1-4 16-KiB chunks in flight, with 4 or 8 waves,
// (b) into VGPRs with plain 16-byte loads.  All workgroups read the same 2-MiB region (an activation / weight panel shared
// through L2).   hipcc --offload-arch=gfx950 -O3 l2fill_probe.hip -o l2fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int REGION_CHUNKS = 128;        // 128 x 16 KiB = 2 MiB

template <int NT, int DEPTH>   // NT threads; DEPTH 16-KiB chunks in flight (ring of 8 chunks = 128 KiB LDS)
__global__ __launch_bounds__(NT) void k_dma(const uint4* __restrict__ src, float* __restrict__ out, int chunks) {
  __shared__ __attribute__((aligned(16))) uint4 buf[8 * 1024];
  const int wave = threadIdx.x >> 6;
  constexpr int PER = 1024 / NT;          // glds per thread per chunk (4 at 256 threads, 2 at 512)
  float acc = 0.f;
  auto issue = [&](int c) {
    const uint4* p = src + (size_t)((c + blockIdx.x * 7) % REGION_CHUNKS) * 1024;
#pragma unroll
    for (int i = 0; i < PER; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i * NT + threadIdx.x),
                                       (__attribute__((address_space(3))) void*)(buf + (c & 7) * 1024 + i * NT + wave * 64), 16, 0, 0);
  };
  for (int c = 0; c < DEPTH && c < chunks; ++c) issue(c);
  for (int c = 0; c < chunks; ++c) {
    // counted wait: leave DEPTH-1 chunks in flight
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
    unsigned t;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((unsigned)(((c & 7) * 1024 + wave * 64 + (threadIdx.x & 63)) * 16)) : "memory");
    acc += __uint_as_float(t);
    issue(c + DEPTH);                     // keeps the in-flight count constant (runs DEPTH chunks past the end: harmless)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 1234.5f) out[0] = acc;
}

// The GEMM kernels' operand pattern: a 16-KiB chunk is 128 rows x 128 B of a row-major matrix with `stride` bytes per row
// (8 rows per wave-instruction: eight separate 128-B lines instead of 1 KiB contiguous), walking along the row (k).
template <int NT, int DEPTH>
__global__ __launch_bounds__(NT) void k_dma_rows(const char* __restrict__ src, float* __restrict__ out, int chunks, int stride, int kchunks,
                                                 int rowblocks) {
  __shared__ __attribute__((aligned(16))) uint4 buf[8 * 1024];
  const int wave = threadIdx.x >> 6;
  constexpr int PER = 1024 / NT;
  float acc = 0.f;
  const int rb = (blockIdx.x * 5) % rowblocks;          // this workgroup's block of 128 rows
  auto issue = [&](int c) {
    const int kc = c % kchunks;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int p = i * NT + threadIdx.x, row = p >> 3, sl = p & 7;
      const char* g = src + ((size_t)(rb * 128 + row)) * stride + (size_t)kc * 128 + sl * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(buf + (c & 7) * 1024 + i * NT + wave * 64), 16, 0, 0);
    }
  };
  for (int c = 0; c < DEPTH && c < chunks; ++c) issue(c);
  for (int c = 0; c < chunks; ++c) {
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER) : "memory");
    unsigned t;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((unsigned)(((c & 7) * 1024 + wave * 64 + (threadIdx.x & 63)) * 16)) : "memory");
    acc += __uint_as_float(t);
    issue(c + DEPTH);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 1234.5f) out[0] = acc;
}

template <int NT>
__global__ __launch_bounds__(NT) void k_vgpr(const uint4* __restrict__ src, float* __restrict__ out, int chunks) {
  float acc = 0.f;
  constexpr int PER = 1024 / NT;
  for (int c = 0; c < chunks; c += 2) {   // 2 chunks = 8 (256 thr) / 4 (512 thr) 16-byte loads in flight per lane
    uint4 v[2 * PER];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint4* p = src + (size_t)((c + u + blockIdx.x * 7) % REGION_CHUNKS) * 1024;
#pragma unroll
      for (int i = 0; i < PER; ++i) v[u * PER + i] = p[i * NT + threadIdx.x];
    }
#pragma unroll
    for (int i = 0; i < 2 * PER; ++i) acc += __uint_as_float(v[i].x ^ v[i].y ^ v[i].z ^ v[i].w);
  }
  if (acc == 1234.5f) out[0] = acc;
}

int main() {
  uint4* src; float* out;
  CHECK(hipMalloc(&src, (size_t)REGION_CHUNKS * 16384)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(src, 1, (size_t)REGION_CHUNKS * 16384));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int chunks = 2048;                // 32 MiB per workgroup
  for (int grid : {256, 512}) {
    auto run = [&](const char* name, auto launch) {
      launch(); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < 3; ++i) launch();
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = 3.0 * grid * chunks * 16384.0;
      printf("%-34s grid %4d: %7.1f GB/s per workgroup, %6.2f TB/s chip\n", name, grid, bytes / grid / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
    };
    run("lds-dma 4 waves, 1 chunk in flight", [&] { hipLaunchKernelGGL((k_dma<256, 1>), dim3(grid), dim3(256), 0, 0, src, out, chunks); });
    run("lds-dma 4 waves, 2 chunks", [&] { hipLaunchKernelGGL((k_dma<256, 2>), dim3(grid), dim3(256), 0, 0, src, out, chunks); });
    run("lds-dma 4 waves, 3 chunks", [&] { hipLaunchKernelGGL((k_dma<256, 3>), dim3(grid), dim3(256), 0, 0, src, out, chunks); });
    run("lds-dma 4 waves, 4 chunks", [&] { hipLaunchKernelGGL((k_dma<256, 4>), dim3(grid), dim3(256), 0, 0, src, out, chunks); });
    run("lds-dma 8 waves, 2 chunks", [&] { hipLaunchKernelGGL((k_dma<512, 2>), dim3(grid), dim3(512), 0, 0, src, out, chunks); });
    run("lds-dma 8 waves, 4 chunks", [&] { hipLaunchKernelGGL((k_dma<512, 4>), dim3(grid), dim3(512), 0, 0, src, out, chunks); });
    {   // row-major matrices as the decode GEMMs see them: 1024 x 1280 and 1024 x 5120 halves (2.6 MB / 10.5 MB)
      char* mat; CHECK(hipMalloc(&mat, (size_t)1024 * 10240 + 65536)); CHECK(hipMemset(mat, 1, (size_t)1024 * 10240 + 65536));
      run("lds-dma rows stride 2560, 1 chunk", [&] { hipLaunchKernelGGL((k_dma_rows<256, 1>), dim3(grid), dim3(256), 0, 0, mat, out, chunks, 2560, 20, 8); });
      run("lds-dma rows stride 2560, 2 chunks", [&] { hipLaunchKernelGGL((k_dma_rows<256, 2>), dim3(grid), dim3(256), 0, 0, mat, out, chunks, 2560, 20, 8); });
      run("lds-dma rows stride 2560, 4 chunks", [&] { hipLaunchKernelGGL((k_dma_rows<256, 4>), dim3(grid), dim3(256), 0, 0, mat, out, chunks, 2560, 20, 8); });
      run("lds-dma rows stride 10240, 1 chunk", [&] { hipLaunchKernelGGL((k_dma_rows<256, 1>), dim3(grid), dim3(256), 0, 0, mat, out, chunks, 10240, 80, 8); });
      run("lds-dma rows stride 10240, 2 chunks", [&] { hipLaunchKernelGGL((k_dma_rows<256, 2>), dim3(grid), dim3(256), 0, 0, mat, out, chunks, 10240, 80, 8); });
      run("lds-dma rows stride 10240, 4 chunks", [&] { hipLaunchKernelGGL((k_dma_rows<256, 4>), dim3(grid), dim3(256), 0, 0, mat, out, chunks, 10240, 80, 8); });
      CHECK(hipFree(mat));
    }
    run("vgpr 16-B loads, 4 waves", [&] { hipLaunchKernelGGL((k_vgpr<256>), dim3(grid), dim3(256), 0, 0, src, out, chunks); });
    run("vgpr 16-B loads, 8 waves", [&] { hipLaunchKernelGGL((k_vgpr<512>), dim3(grid), dim3(512), 0, 0, src, out, chunks); });
  }
  return 0;
}
