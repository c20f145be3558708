// Which compute units does a stream made with hipExtStreamCreateWithCUMask really use on MI355X (8 XCDs x 32 CUs)?
// Every workgroup records (XCC_ID, HW_ID); the host counts distinct (xcc, se, sh, cu) per mask pattern.
//   hipcc --offload-arch=gfx950 -O2 -o cu_mask cu_mask.hip && ./cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#include <map>
__global__ void k_where(uint32_t* out, unsigned long long spin) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("CUs %d\n", cus);
  const int n_wg = 8192;
  uint32_t* d;
  hipMalloc(&d, n_wg * 8);
  std::vector<uint32_t> h(2 * n_wg);
  for (int pattern = 0; pattern < 6; ++pattern) {
    std::vector<uint32_t> mask((cus + 31) / 32, 0u);
    const char* name = "";
    for (int i = 0; i < cus; ++i) {
      bool on = true;
      switch (pattern) {
        case 0: name = "all"; break;
        case 1: name = "every 8th off"; on = i % 8 != 7; break;
        case 2: name = "every 16th off"; on = i % 16 != 15; break;
        case 3: name = "first 128 bits on"; on = i < 128; break;
        case 4: name = "bits 0..31 on"; on = i < 32; break;
        case 5: name = "even bits on"; on = i % 2 == 0; break;
      }
      if (on) mask[i / 32] |= 1u << (i % 32);
    }
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); continue; }
    hipLaunchKernelGGL(k_where, dim3(n_wg), dim3(256), 0, st, d, 200000ull);
    hipStreamSynchronize(st);
    hipMemcpy(h.data(), d, n_wg * 8, hipMemcpyDeviceToHost);
    std::set<uint32_t> seen;
    std::map<uint32_t, int> per_xcc;
    for (int i = 0; i < n_wg; ++i) {
      const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
      const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      const uint32_t key = (xcc << 16) | (se << 8) | (sh << 4) | cu;
      if (seen.insert(key).second) per_xcc[xcc]++;
    }
    printf("%-20s distinct CUs used %3zu; per XCC:", name, seen.size());
    for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
    hipStreamDestroy(st);
  }
  return 0;
}
