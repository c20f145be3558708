// merlin_dev.hpp -- Keccak-f[1600] on the device, one state per lane in registers
// (24 rolled rounds).  The STROBE / Merlin framing around it is data, not code: the
// per-shape tape of transcript_tape.hpp, run by k_transcript (prep_kernels.hpp).
// (FIPS 202.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zk {

__device__ __forceinline__ uint64_t rotl64_dev(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

__device__ inline void keccak_f1600_regs(uint64_t a[25]) {
  const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#pragma unroll 1
  for (int rnd = 0; rnd < 24; ++rnd) {
    uint64_t c[5], d[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl64_dev(c[(x + 1) % 5], 1);
#pragma unroll
    for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
    // rho + pi
    uint64_t b[25];
    constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
      for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64_dev(a[x + 5 * y], RHO[x + 5 * y]);
    // chi
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
      for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= RC[rnd];
  }
}

// The same permutation on explicit 32-bit halves: rotations are two v_alignbit_b32 (the 64-bit shifts
// the compiler emits for rotl64 are slower and need an OR each), five-way XORs pair up into v_xor3_b32.
__device__ __forceinline__ void rot64_halves(uint32_t& lo, uint32_t& hi, uint32_t a_lo, uint32_t a_hi, int n) {
  if (n >= 32) { const uint32_t t = a_lo; a_lo = a_hi; a_hi = t; n -= 32; }
  if (n == 0) { lo = a_lo; hi = a_hi; return; }
  lo = __builtin_amdgcn_alignbit(a_lo, a_hi, 32 - n);
  hi = __builtin_amdgcn_alignbit(a_hi, a_lo, 32 - n);
}

__device__ inline void keccak_f1600_halves(uint32_t lo[25], uint32_t hi[25]) {
  const uint32_t RC_LO[24] = {0x00000001u, 0x00008082u, 0x0000808Au, 0x80008000u, 0x0000808Bu, 0x80000001u, 0x80008081u, 0x00008009u,
                              0x0000008Au, 0x00000088u, 0x80008009u, 0x8000000Au, 0x8000808Bu, 0x0000008Bu, 0x00008089u, 0x00008003u,
                              0x00008002u, 0x00000080u, 0x0000800Au, 0x8000000Au, 0x80008081u, 0x00008080u, 0x80000001u, 0x80008008u};
  const uint32_t RC_HI[24] = {0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u,
                              0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u,
                              0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u};
#pragma unroll 1
  for (int rnd = 0; rnd < 24; ++rnd) {
    uint32_t cl[5], ch[5], rl[5], rh[5];
#pragma unroll
    for (int x = 0; x < 5; ++x) {   // column parities, three inputs per v_bitop3_b32 (0x96 = a ^ b ^ c)
      cl[x] = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(lo[x], lo[x + 5], lo[x + 10], 0x96), lo[x + 15], lo[x + 20], 0x96);
      ch[x] = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_bitop3_b32(hi[x], hi[x + 5], hi[x + 10], 0x96), hi[x + 15], hi[x + 20], 0x96);
    }
#pragma unroll
    for (int x = 0; x < 5; ++x) rot64_halves(rl[x], rh[x], cl[x], ch[x], 1);
    uint32_t bl[25], bh[25];
    constexpr int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
      for (int y = 0; y < 5; ++y) {   // theta (a ^ C[x-1] ^ rot(C[x+1], 1) in one instruction), rho, pi
        const int i = x + 5 * y;
        const uint32_t tl = __builtin_amdgcn_bitop3_b32(lo[i], cl[(x + 4) % 5], rl[(x + 1) % 5], 0x96);
        const uint32_t th = __builtin_amdgcn_bitop3_b32(hi[i], ch[(x + 4) % 5], rh[(x + 1) % 5], 0x96);
        rot64_halves(bl[y + 5 * ((2 * x + 3 * y) % 5)], bh[y + 5 * ((2 * x + 3 * y) % 5)], tl, th, RHO[i]);
      }
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
      for (int x = 0; x < 5; ++x) {
        lo[x + 5 * y] = bl[x + 5 * y] ^ (~bl[(x + 1) % 5 + 5 * y] & bl[(x + 2) % 5 + 5 * y]);
        hi[x + 5 * y] = bh[x + 5 * y] ^ (~bh[(x + 1) % 5 + 5 * y] & bh[(x + 2) % 5 + 5 * y]);
      }
    lo[0] ^= RC_LO[rnd];
    hi[0] ^= RC_HI[rnd];
  }
}

}  // namespace zk
