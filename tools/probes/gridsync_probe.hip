// Cost of a device-wide barrier among one workgroup per CU on gfx950 (cooperative launch): cooperative-groups grid.sync()
// and a hand-rolled sense-reversing barrier (agent-scope atomics + fences).   hipcc --offload-arch=gfx950 -O3 gridsync_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void k_cg(int n, float* out) {
  cg::grid_group g = cg::this_grid();
  float a = threadIdx.x;
  for (int i = 0; i < n; ++i) { a = a * 1.0001f + 1.f; g.sync(); }
  if (a == 12345.f) out[0] = a;
}

// bar[0]: arrival counter, bar[1]: generation
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned nblocks, unsigned& gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();                                        // release: this workgroup's writes are visible device-wide
    const unsigned target = gen + 1;
    if (atomicAdd(&bar[0], 1u) == nblocks - 1) {
      bar[0] = 0;
      __threadfence();
      atomicExch(&bar[1], target);
    } else {
      long spins = 0;
      while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 50000000) break;                      // bounded: never hang the box
      }
    }
    __threadfence();                                        // acquire
  }
  gen += 1;
  __syncthreads();
}

// two-level: one counter per XCD (workgroup id % 8, 128 bytes apart), the last arriver of each XCD goes to the top counter
__device__ __forceinline__ void grid_barrier2(unsigned* bar, unsigned nblocks, unsigned& gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned target = gen + 1, xcd = blockIdx.x & 7, per = nblocks >> 3;
    unsigned* cx = bar + 64 + xcd * 32;
    if (atomicAdd(cx, 1u) == per - 1) {
      *cx = 0;
      if (atomicAdd(&bar[0], 1u) == 7) { bar[0] = 0; __threadfence(); atomicExch(&bar[1], target); }
    }
    long spins = 0;
    while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > 50000000) break;
    }
    __threadfence();
  }
  gen += 1;
  __syncthreads();
}

__global__ void k_own2(int n, unsigned* bar, float* out) {
  unsigned gen = 0;
  float a = threadIdx.x;
  for (int i = 0; i < n; ++i) { a = a * 1.0001f + 1.f; grid_barrier2(bar, gridDim.x, gen); }
  if (a == 12345.f) out[0] = a;
}

__global__ void k_own(int n, unsigned* bar, float* out) {
  unsigned gen = 0;
  float a = threadIdx.x;
  for (int i = 0; i < n; ++i) { a = a * 1.0001f + 1.f; grid_barrier(bar, gridDim.x, gen); }
  if (a == 12345.f) out[0] = a;
}

int main() {
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; unsigned* bar;
  hipMalloc(&out, 4); hipMalloc(&bar, 4096); hipMemset(bar, 0, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    int n = 2000, grid = ncu * wgs_per_cu;
    void* args1[] = {&n, &out};
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipError_t e = hipLaunchCooperativeKernel((void*)k_cg, dim3(grid), dim3(256), args1, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("cg grid.sync   : %d workgroups: %.2f us per barrier (%s)\n", grid, ms * 1e3 / n, hipGetErrorString(e));
    }
    void* args2[] = {&n, &bar, &out};
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(bar, 0, 4096);
      hipEventRecord(e0);
      hipError_t e = hipLaunchCooperativeKernel((void*)k_own, dim3(grid), dim3(256), args2, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("own barrier    : %d workgroups: %.2f us per barrier (%s)\n", grid, ms * 1e3 / n, hipGetErrorString(e));
    }
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(bar, 0, 4096);
      hipEventRecord(e0);
      hipError_t e = hipLaunchCooperativeKernel((void*)k_own2, dim3(grid), dim3(256), args2, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("two-level      : %d workgroups: %.2f us per barrier (%s)\n", grid, ms * 1e3 / n, hipGetErrorString(e));
    }
  }
  return 0;
}
