// Micro-benchmark: issue cost of the individual VALU opcodes that the field / scalar / point kernels of this library are
// made of on gfx950, one opcode per kernel through inline assembly (the compiler cannot substitute another instruction),
// eight independent chains per lane, WAVES_PER_SIMD wavefronts per SIMD (so that dependent-issue latency is covered and the
// figure is the ISSUE cost: cycles one SIMD spends per wave-instruction).  Feeds tools/isa_mix.py: histogram of a kernel's
// opcodes x these costs = the mix-weighted bound of the kernel ("integer-multiply
// rate; say which binds").
//
// No asm statement declares a clobber: with "vcc" declared the compiler put an s_nop after EVERY statement, and a lone wavefront
// then read 8.2 cycles per instruction (VALU + nop) where it issues every ~5.  Opcodes that write a scalar pair use s[20:21],
// which the kernels do not otherwise touch (checked in the assembly).
// Build: hipcc -O3 --offload-arch=gfx950 valu_ops.hip -o valu_ops     Output: one line per opcode, `name cycles`.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int CH = 8;            // independent chains per lane
constexpr int REP = 16;          // the CH statements are repeated REP times per loop iteration: a taken branch costs ~50 cycles, which a
                                 // lone wavefront cannot hide (with 8 statements per iteration it read 11 cycles per instruction)

#define KERNEL32(NAME, ASMTEXT)                                                          \
  __global__ void NAME(uint32_t* out, uint32_t a0, int iters) {                          \
    uint32_t x[CH], a = a0 + threadIdx.x, b = a0 * 3 + 1;                                \
    _Pragma("unroll") for (int j = 0; j < CH; ++j) x[j] = a + j;                         \
    for (int i = 0; i < iters; ++i) {                                                    \
      _Pragma("unroll") for (int r = 0; r < REP; ++r)                                    \
      _Pragma("unroll") for (int j = 0; j < CH; ++j)                                     \
        asm volatile(ASMTEXT : "+v"(x[j]) : "v"(a), "v"(b));              \
    }                                                                                    \
    uint32_t s = 0;                                                                      \
    _Pragma("unroll") for (int j = 0; j < CH; ++j) s ^= x[j];                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
  }

#define KERNEL64(NAME, ASMTEXT)                                                          \
  __global__ void NAME(uint32_t* out, uint32_t a0, int iters) {                          \
    uint64_t x[CH];                                                                      \
    uint32_t a = a0 + threadIdx.x, b = a0 * 3 + 1;                                       \
    uint64_t w = ((uint64_t)a << 32) | b;                                                \
    _Pragma("unroll") for (int j = 0; j < CH; ++j) x[j] = w + j;                         \
    for (int i = 0; i < iters; ++i) {                                                    \
      _Pragma("unroll") for (int r = 0; r < REP; ++r)                                    \
      _Pragma("unroll") for (int j = 0; j < CH; ++j)                                     \
        asm volatile(ASMTEXT : "+v"(x[j]) : "v"(a), "v"(b), "v"(w));      \
    }                                                                                    \
    uint64_t s = 0;                                                                      \
    _Pragma("unroll") for (int j = 0; j < CH; ++j) s ^= x[j];                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(s ^ (s >> 32));             \
  }

// 32-bit
KERNEL32(k_v_add_u32, "v_add_u32 %0, %1, %0")
KERNEL32(k_v_sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL32(k_v_and_b32, "v_and_b32 %0, %1, %0")
KERNEL32(k_v_or_b32, "v_or_b32 %0, %1, %0")
KERNEL32(k_v_xor_b32, "v_xor_b32 %0, %1, %0")
KERNEL32(k_v_mov_b32, "v_mov_b32 %0, %1")
KERNEL32(k_v_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
KERNEL32(k_v_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
KERNEL32(k_v_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
KERNEL32(k_v_add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 3")
KERNEL32(k_v_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL32(k_v_and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_v_add3_u32, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_v_alignbit_b32, "v_alignbit_b32 %0, %0, %1, 13")
KERNEL32(k_v_bfe_u32, "v_bfe_u32 %0, %0, 3, 26")
KERNEL32(k_v_bfi_b32, "v_bfi_b32 %0, %1, %0, %2")
KERNEL32(k_v_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_v_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_v_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_v_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_v_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_v_add_co_u32, "v_add_co_u32 %0, s[20:21], %0, %1")
KERNEL32(k_v_addc_co_u32, "v_addc_co_u32 %0, s[20:21], %0, %1, s[20:21]")
KERNEL32(k_v_subb_co_u32, "v_subb_co_u32 %0, s[20:21], %0, %1, s[20:21]")
KERNEL32(k_v_cmp_lt_u32, "v_cmp_lt_u32_e64 s[20:21], %0, %1")
KERNEL32(k_v_mov_b32_dpp_quad, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL32(k_v_add_u32_dpp_quad, "v_add_u32_dpp %0, %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
KERNEL32(k_v_mov_b32_dpp_row_ror, "v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf")
KERNEL32(k_v_readfirstlane_b32, "v_readfirstlane_b32 s20, %0")
KERNEL32(k_v_readlane_b32, "v_readlane_b32 s20, %0, 5")
// 64-bit (register pairs)
KERNEL64(k_v_lshrrev_b64, "v_lshrrev_b64 %0, 13, %0")
KERNEL64(k_v_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
KERNEL64(k_v_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %3")
KERNEL64(k_v_mad_u64_u32, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0")
KERNEL64(k_v_mad_i64_i32, "v_mad_i64_i32 %0, s[20:21], %1, %2, %0")
KERNEL64(k_v_add_f64, "v_add_f64 %0, %0, %3")
KERNEL64(k_v_fma_f64, "v_fma_f64 %0, %0, %3, %3")
KERNEL64(k_v_mov_b64, "v_mov_b64 %0, %3")
KERNEL64(k_v_pk_add_u16_as_pair, "v_pk_mov_b32 %0, %3, %3")

// conditional moves, by the forms the compiler emits and the hazards around them (the generic macro's "vcc" clobber makes the
// compiler put an s_nop between two of them; these kernels clobber nothing)
#define KERNEL32_NOCLOB(NAME, ASMTEXT)                                                   \
  __global__ void NAME(uint32_t* out, uint32_t a0, int iters) {                          \
    uint32_t x[CH], a = a0 + threadIdx.x, b = a0 * 3 + 1;                                \
    _Pragma("unroll") for (int j = 0; j < CH; ++j) x[j] = a + j;                         \
    for (int i = 0; i < iters; ++i) {                                                    \
      _Pragma("unroll") for (int r = 0; r < REP; ++r)                                    \
      _Pragma("unroll") for (int j = 0; j < CH; ++j)                                     \
        asm volatile(ASMTEXT : "+v"(x[j]) : "v"(a), "v"(b));                             \
    }                                                                                    \
    uint32_t s = 0;                                                                      \
    _Pragma("unroll") for (int j = 0; j < CH; ++j) s ^= x[j];                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
  }
KERNEL32_NOCLOB(k_v_cndmask_b32_e32_vcc, "v_cndmask_b32_e32 %0, %0, %1, vcc")
KERNEL32_NOCLOB(k_v_cndmask_b32_e64_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32_NOCLOB(k_cmp_then_cndmask_pair, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]")
KERNEL32_NOCLOB(k_cmp_then_10_cndmask, "v_cmp_lt_u32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL32_NOCLOB(k_cmp_vcc_then_cndmask_vcc, "v_cmp_lt_u32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc")
KERNEL32_NOCLOB(k_cmp_vcc_then_4_cndmask_vcc, "v_cmp_lt_u32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc\n\tv_cndmask_b32_e32 %0, %0, %2, vcc\n\tv_cndmask_b32_e32 %0, %0, %1, vcc")
KERNEL32_NOCLOB(k_cmp_vcc_nop_cndmask_vcc, "v_cmp_lt_u32_e32 vcc, %0, %1\n\ts_nop 1\n\tv_cndmask_b32_e32 %0, %0, %2, vcc")
KERNEL32_NOCLOB(k_cmp_vcc_only, "v_cmp_lt_u32_e32 vcc, %0, %1")
KERNEL32_NOCLOB(k_cmp_sgpr_only, "v_cmp_lt_u32_e64 s[20:21], %0, %1")
KERNEL32_NOCLOB(k_select_by_xor_and_xor, "v_xor_b32 %0, %0, %1\n\tv_and_b32 %0, %0, %2\n\tv_xor_b32 %0, %0, %1")
KERNEL32_NOCLOB(k_s_nop_0_x1, "s_nop 0")

struct Entry { const char* name; void (*fn)(uint32_t*, uint32_t, int); };
#define E(k) {#k, k}

int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 4;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  printf("# device: CUs=%d clock=%.3f GHz waves_per_SIMD=%d chains_per_lane=%d\n", cus, ghz, waves_per_simd, CH);
  printf("# opcode  cycles_per_wave_instruction_per_SIMD  (one SIMD issues one VALU instruction at a time)\n");
  // default: 256 threads per block (4 waves), blocks per CU = waves per SIMD.  argv[2] = threads per block (64: one wave per
  // block, 4 x as many blocks): where a workgroup's four waves land is the dispatcher's business -- with 256 the "w=1" column
  // reads 8.2 cycles, with 64-thread blocks it shows what a wave alone on a SIMD really issues at
  const int threads = argc > 2 ? atoi(argv[2]) : 256, blocks = cus * waves_per_simd * (256 / threads);
  uint32_t* out = nullptr;
  CK(hipMalloc(&out, (size_t)blocks * threads * 4));
  const Entry table[] = {
    E(k_v_add_u32), E(k_v_sub_u32), E(k_v_and_b32), E(k_v_or_b32), E(k_v_xor_b32), E(k_v_mov_b32), E(k_v_lshlrev_b32), E(k_v_lshrrev_b32),
    E(k_v_lshl_add_u32), E(k_v_add_lshl_u32), E(k_v_lshl_or_b32), E(k_v_and_or_b32), E(k_v_add3_u32), E(k_v_alignbit_b32), E(k_v_bfe_u32),
    E(k_v_bfi_b32), E(k_v_mul_lo_u32), E(k_v_mul_hi_u32), E(k_v_mul_u32_u24), E(k_v_mad_u32_u24), E(k_v_cndmask_b32), E(k_v_add_co_u32),
    E(k_v_addc_co_u32), E(k_v_subb_co_u32), E(k_v_cmp_lt_u32), E(k_v_mov_b32_dpp_quad), E(k_v_add_u32_dpp_quad), E(k_v_mov_b32_dpp_row_ror),
    E(k_v_readfirstlane_b32), E(k_v_readlane_b32), E(k_v_lshrrev_b64), E(k_v_lshlrev_b64), E(k_v_lshl_add_u64), E(k_v_mad_u64_u32),
    E(k_v_mad_i64_i32), E(k_v_add_f64), E(k_v_fma_f64), E(k_v_mov_b64), E(k_v_pk_add_u16_as_pair),
    E(k_v_cndmask_b32_e32_vcc), E(k_v_cndmask_b32_e64_sgpr), E(k_cmp_then_cndmask_pair), E(k_cmp_then_10_cndmask), E(k_cmp_vcc_then_cndmask_vcc), E(k_cmp_vcc_then_4_cndmask_vcc),
    E(k_cmp_vcc_nop_cndmask_vcc), E(k_cmp_vcc_only), E(k_cmp_sgpr_only), E(k_select_by_xor_and_xor), E(k_s_nop_0_x1),
  };
  hipEvent_t t0, t1;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  for (const Entry& e : table) {
    const char* shown = !strcmp(e.name, "k_v_pk_add_u16_as_pair") ? "v_pk_mov_b32" : e.name + 2;
    // two lengths, best of five each: the difference is free of launch, ramp and drain
    const int n1 = 512, n2 = 2560;
    float best[2] = {1e30f, 1e30f};
    hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, out, 7u, 256);      // warm
    CK(hipDeviceSynchronize());
    for (int which = 0; which < 2; ++which)
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(t0, 0));
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(threads), 0, 0, out, 7u + rep, which ? n2 : n1);
        CK(hipEventRecord(t1, 0));
        CK(hipEventSynchronize(t1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, t0, t1));
        if (ms < best[which]) best[which] = ms;
      }
    // wave-instructions per SIMD in the difference = waves_per_simd * (n2 - n1) * CH * REP; cycles = time * clock
    const double insts_per_simd = (double)waves_per_simd * (n2 - n1) * CH * REP;
    printf("%-26s %7.3f   (%.3f ms - %.3f ms)\n", shown, (best[1] - best[0]) * 1e-3 * ghz * 1e9 / insts_per_simd, best[1], best[0]);
  }
  CK(hipFree(out));
  return 0;
}
