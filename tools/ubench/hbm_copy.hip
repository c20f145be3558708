// hbm_copy.hip -- which shape of a plain copy kernel reaches the achievable HBM rate on MI355X (the yardstick of
// zkgpu_measure_hbm_copy; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).  Variants: grid-stride with 4 loads in
// flight (the library's kernel until round 3), block-contiguous tiles, non-temporal loads / stores, 8 loads in flight.
//   hipcc -O3 --offload-arch=gfx950 -o hbm_copy tools/ubench/hbm_copy.hip && ./hbm_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_stride4(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const u32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}

// every workgroup owns tiles of UNROLL x 256 consecutive vectors (UNROLL x 4 KiB), tile after tile in grid order
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_tile(const u32x4* __restrict__ src, u32x4* __restrict__ dst, uint64_t n) {
  const uint64_t tile = (uint64_t)UNROLL * 256;
  for (uint64_t base = (uint64_t)blockIdx.x * tile; base < n; base += (uint64_t)gridDim.x * tile) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint64_t i = base + (uint64_t)u * 256 + threadIdx.x;
      if (i < n) v[u] = NT ? __builtin_nontemporal_load(&src[i]) : src[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const uint64_t i = base + (uint64_t)u * 256 + threadIdx.x;
      if (i < n) { if (NT) __builtin_nontemporal_store(v[u], &dst[i]); else dst[i] = v[u]; }
    }
  }
}

template <typename F>
static double time_it(F launch, size_t bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(e0, 0);
    launch(it);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  return 2.0 * bytes / (best * 1e-3) / 1e9;
}

int main() {
  const size_t bytes = (size_t)2 << 30;
  void *a, *b;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 0x5a, bytes);
  const uint64_t n = bytes / 16;
  for (unsigned blocks : {256u * 4, 256u * 8, 256u * 16, 256u * 32, 256u * 64}) {
    printf("blocks %6u: stride4 %7.0f", blocks, time_it([&](int it) { hipLaunchKernelGGL(k_stride4, dim3(blocks), dim3(256), 0, 0, (const u32x4*)(it & 1 ? b : a), (u32x4*)(it & 1 ? a : b), n); }, bytes));
    printf("  tile4 %7.0f", time_it([&](int it) { hipLaunchKernelGGL((k_tile<4, false>), dim3(blocks), dim3(256), 0, 0, (const u32x4*)(it & 1 ? b : a), (u32x4*)(it & 1 ? a : b), n); }, bytes));
    printf("  tile4nt %7.0f", time_it([&](int it) { hipLaunchKernelGGL((k_tile<4, true>), dim3(blocks), dim3(256), 0, 0, (const u32x4*)(it & 1 ? b : a), (u32x4*)(it & 1 ? a : b), n); }, bytes));
    printf("  tile8 %7.0f", time_it([&](int it) { hipLaunchKernelGGL((k_tile<8, false>), dim3(blocks), dim3(256), 0, 0, (const u32x4*)(it & 1 ? b : a), (u32x4*)(it & 1 ? a : b), n); }, bytes));
    printf("  tile8nt %7.0f", time_it([&](int it) { hipLaunchKernelGGL((k_tile<8, true>), dim3(blocks), dim3(256), 0, 0, (const u32x4*)(it & 1 ? b : a), (u32x4*)(it & 1 ? a : b), n); }, bytes));
    printf("  tile2nt %7.0f GB/s\n", time_it([&](int it) { hipLaunchKernelGGL((k_tile<2, true>), dim3(blocks), dim3(256), 0, 0, (const u32x4*)(it & 1 ? b : a), (u32x4*)(it & 1 ? a : b), n); }, bytes));
  }
  // one tile per workgroup, no loop at all (grid = n / tile)
  printf("one tile per workgroup: tile4nt %7.0f", time_it([&](int it) { hipLaunchKernelGGL((k_tile<4, true>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, (const u32x4*)(it & 1 ? b : a), (u32x4*)(it & 1 ? a : b), n); }, bytes));
  printf("  tile4 %7.0f", time_it([&](int it) { hipLaunchKernelGGL((k_tile<4, false>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, (const u32x4*)(it & 1 ? b : a), (u32x4*)(it & 1 ? a : b), n); }, bytes));
  printf("  tile1 %7.0f", time_it([&](int it) { hipLaunchKernelGGL((k_tile<1, false>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, (const u32x4*)(it & 1 ? b : a), (u32x4*)(it & 1 ? a : b), n); }, bytes));
  printf("  hipMemcpyDtoD %7.0f GB/s\n", time_it([&](int it) { hipMemcpyAsync(it & 1 ? a : b, it & 1 ? b : a, bytes, hipMemcpyDeviceToDevice, 0); }, bytes));
  return 0;
}
