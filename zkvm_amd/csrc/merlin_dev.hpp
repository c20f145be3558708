// merlin_dev.hpp -- Keccak-f[1600] / STROBE-128 / Merlin on the device.
//
// The Fiat-Shamir transcript of r1cs::Verifier::verify is strictly sequential
// per proof, so it runs one lane per transaction.  The 200-byte STROBE state
// lives in LDS as 50 words per lane, word-interleaved over the lanes of the
// block (address = word * blockDim + lane: conflict-free), because STROBE
// touches it at byte positions that are only known at run time; Keccak-f pulls
// it into registers, runs 24 rolled rounds, and puts it back.  All lanes of a
// launch replay transcripts of the same shape, so every position is
// wave-uniform and there is no divergence.
// (merlin.cool, STROBE v1.0.2, FIPS 202.)
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

// STROBE-128 state of one lane inside a block-wide LDS array
struct StrobeDev {
  uint32_t* st;        // &lds[lane]; word w of this lane is st[w * stride]
  uint32_t stride;
  uint32_t pos, pos_begin;

  static constexpr uint32_t R = 166;
  __device__ __forceinline__ void xor_byte(uint32_t i, uint32_t b) { st[(i >> 2) * stride] ^= b << (8 * (i & 3)); }
  __device__ __forceinline__ uint32_t get_byte(uint32_t i) const { return (st[(i >> 2) * stride] >> (8 * (i & 3))) & 0xff; }
  __device__ __forceinline__ void clear_byte(uint32_t i) { st[(i >> 2) * stride] &= ~(0xffu << (8 * (i & 3))); }

  __device__ inline void permute() {
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) a[i] = (uint64_t)st[(2 * i) * stride] | ((uint64_t)st[(2 * i + 1) * stride] << 32);
    keccak_f1600_regs(a);
#pragma unroll
    for (int i = 0; i < 25; ++i) { st[(2 * i) * stride] = (uint32_t)a[i]; st[(2 * i + 1) * stride] = (uint32_t)(a[i] >> 32); }
  }
  __device__ inline void run_f() {
    xor_byte(pos, pos_begin);
    xor_byte(pos + 1, 0x04);
    xor_byte(R + 1, 0x80);
    permute();
    pos = 0;
    pos_begin = 0;
  }
  __device__ inline void absorb_byte(uint32_t b) {
    xor_byte(pos, b);
    if (++pos == R) run_f();
  }
  __device__ inline void begin_op(uint32_t flags) {
    const uint32_t old_begin = pos_begin;
    pos_begin = pos + 1;
    absorb_byte(old_begin);
    absorb_byte(flags);
    if ((flags & (4u | 32u)) && pos != 0) run_f();
  }
  // uniform bytes (labels, lengths)
  __device__ inline void absorb_const(const char* s, uint32_t n) { for (uint32_t i = 0; i < n; ++i) absorb_byte((uint8_t)s[i]); }
  // per-lane little-endian words
  __device__ inline void absorb_words(const uint32_t* w, uint32_t nwords) {
    for (uint32_t i = 0; i < nwords; ++i) {
      const uint32_t v = w[i];
      absorb_byte(v & 0xff); absorb_byte((v >> 8) & 0xff); absorb_byte((v >> 16) & 0xff); absorb_byte(v >> 24);
    }
  }
  __device__ inline void le32(uint32_t n) { absorb_byte(n & 0xff); absorb_byte((n >> 8) & 0xff); absorb_byte((n >> 16) & 0xff); absorb_byte(n >> 24); }

  // Merlin framing
  __device__ inline void append_message_words(const char* label, uint32_t label_len, const uint32_t* w, uint32_t nwords) {
    begin_op(16u | 2u); absorb_const(label, label_len);   // meta-AD(label)
    le32(4 * nwords);                                      // meta-AD(len), continuation
    begin_op(2u); absorb_words(w, nwords);                 // AD(data)
  }
  __device__ inline void append_message_const(const char* label, uint32_t label_len, const char* msg, uint32_t n) {
    begin_op(16u | 2u); absorb_const(label, label_len);
    le32(n);
    begin_op(2u); absorb_const(msg, n);
  }
  __device__ inline void append_u64(const char* label, uint32_t label_len, uint64_t x) {
    begin_op(16u | 2u); absorb_const(label, label_len);
    le32(8);
    begin_op(2u);
    for (int i = 0; i < 8; ++i) absorb_byte((uint32_t)(x >> (8 * i)) & 0xff);
  }
  // 64 challenge bytes as 16 words
  __device__ inline void challenge_wide(const char* label, uint32_t label_len, uint32_t out[16]) {
    begin_op(16u | 2u); absorb_const(label, label_len);
    le32(64);
    begin_op(1u | 2u | 4u);                                // PRF
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
      uint32_t v = 0;
      for (int k = 0; k < 4; ++k) {
        v |= get_byte(pos) << (8 * k);
        clear_byte(pos);
        if (++pos == R) run_f();
      }
      out[i] = v;
    }
  }
};

}  // namespace zk
