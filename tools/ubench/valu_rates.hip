// Micro-benchmark: chip-wide throughput of the VALU instructions a
// GF(2^255-19) multiplier can be built from on gfx950.  Decides the device
// field representation (see DESIGN.md "Field arithmetic on CDNA4").
//
// Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;   // independent chains per thread

__global__ void k_mad_u64_u32(uint64_t* out, uint32_t a0, uint32_t b0) {
  uint64_t acc[UNROLL];
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) acc[j] = j;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      acc[j] = (uint64_t)a * (uint32_t)(b + j) + acc[j];
      asm volatile("" : "+v"(acc[j]));
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) s ^= acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mul_lo_hi(uint64_t* out, uint32_t a0, uint32_t b0) {
  uint32_t lo[UNROLL], hi[UNROLL];
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) { lo[j] = j + a; hi[j] = j + b; }
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      uint32_t l = lo[j] * hi[j];
      uint32_t h = __umulhi(lo[j], hi[j]);
      lo[j] = l; hi[j] = h | 1;
      asm volatile("" : "+v"(lo[j]), "+v"(hi[j]));
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) s ^= ((uint64_t)hi[j] << 32) | lo[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mul_u24(uint64_t* out, uint32_t a0, uint32_t b0) {
  uint32_t lo[UNROLL], hi[UNROLL];
  uint32_t a = a0 + threadIdx.x, b = b0 + blockIdx.x;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) { lo[j] = j + a; hi[j] = j + b; }
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      uint32_t l = (lo[j] & 0xffffffu) * (hi[j] & 0xffffffu);
      uint32_t h = (uint32_t)(((uint64_t)(lo[j] & 0xffffffu) * (uint64_t)(hi[j] & 0xffffffu)) >> 32);
      lo[j] = l; hi[j] = h | 1;
      asm volatile("" : "+v"(lo[j]), "+v"(hi[j]));
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) s ^= ((uint64_t)hi[j] << 32) | lo[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma_f64(uint64_t* out, double a0, double b0) {
  double acc[UNROLL];
  double a = a0 + threadIdx.x * 1e-9, b = b0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) acc[j] = j;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      acc[j] = __builtin_fma(a, acc[j], b);
      asm volatile("" : "+v"(acc[j]));
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)__double_as_longlong(s);
}

__global__ void k_add_f64(uint64_t* out, double a0, double b0) {
  double acc[UNROLL];
  double a = a0 + threadIdx.x * 1e-9;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + b0;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      acc[j] = acc[j] + a;
      asm volatile("" : "+v"(acc[j]));
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)__double_as_longlong(s);
}

__global__ void k_add_u64(uint64_t* out, uint64_t a0, uint64_t b0) {
  uint64_t acc[UNROLL];
  uint64_t a = a0 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + b0;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      acc[j] = acc[j] + a;
      asm volatile("" : "+v"(acc[j]));
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) s ^= acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_add_u32(uint64_t* out, uint32_t a0, uint32_t b0) {
  uint32_t acc[UNROLL];
  uint32_t a = a0 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) acc[j] = j + b0;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      acc[j] = acc[j] + a;
      asm volatile("" : "+v"(acc[j]));
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) s ^= acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_shr_u64(uint64_t* out, uint64_t a0, uint64_t b0) {
  uint64_t acc[UNROLL];
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) acc[j] = (a0 + j + threadIdx.x) | (b0 << 60);
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      acc[j] = (acc[j] >> 3) | (1ull << 63);
      asm volatile("" : "+v"(acc[j]));
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) s ^= acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma_f32(uint64_t* out, float a0, float b0) {
  float acc[UNROLL];
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) acc[j] = j;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) {
      acc[j] = __builtin_fmaf(a, acc[j], b);
      asm volatile("" : "+v"(acc[j]));
    }
  }
  float s = 0;
#pragma unroll
  for (int j = 0; j < UNROLL; ++j) s += acc[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)__float_as_uint(s);
}

template <typename F>
static int run(const char* name, F launch, double ops_per_thread_iter, int blocks, int threads) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();  // warm
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  double total_ops = (double)blocks * threads * ITERS * UNROLL * ops_per_thread_iter * reps;
  double gops = total_ops / (ms * 1e-3) / 1e9;
  // cycles per wave-instruction per SIMD at 2.4 GHz, 1024 SIMDs
  double inst_per_s_per_simd = gops * 1e9 / 64.0 / 1024.0;
  printf("%-16s %8.3f ms  %10.1f Gop/s  ~%5.2f cyc/wave-inst/SIMD @2.4GHz\n", name, ms / reps, gops,
         2.4e9 / inst_per_s_per_simd);
  return 0;
}

int main() {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  printf("device: %s  CUs=%d  clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  const int threads = 256, blocks = 256 * 8;
  uint64_t* out; CK(hipMalloc(&out, (size_t)blocks * threads * 8));
#define RUN(K, ops, ...) run(#K, [&] { hipLaunchKernelGGL(K, dim3(blocks), dim3(threads), 0, 0, out, __VA_ARGS__); }, ops, blocks, threads)
  RUN(k_fma_f32, 1, 1.0001f, 0.5f);
  RUN(k_add_u32, 1, 3u, 5u);
  RUN(k_add_u64, 1, 3ull, 5ull);
  RUN(k_shr_u64, 1, 3ull, 5ull);
  RUN(k_mad_u64_u32, 1, 3u, 5u);
  RUN(k_mul_lo_hi, 2, 3u, 5u);
  RUN(k_mul_u24, 2, 3u, 5u);
  RUN(k_fma_f64, 1, 1.0000001, 0.5);
  RUN(k_add_f64, 1, 1.0000001, 0.5);
  CK(hipFree(out));
  return 0;
}
