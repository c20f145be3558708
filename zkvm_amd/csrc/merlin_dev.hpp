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

}  // namespace zk
