// keccak.hpp -- Keccak-f[1600] with the SHA3-512 / SHAKE256 framings (FIPS 202),
// host side.  Used for generator derivation (bulletproofs `GeneratorsChain`,
// `PedersenGens::default`) and by the Merlin
// transcript.  Lane-major state (25 x u64), byte access through the lanes.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace zk {

inline uint64_t rotl64(uint64_t x, unsigned n) { return n ? (x << n) | (x >> (64 - n)) : x; }

inline void keccak_f1600(uint64_t s[25]) {
  static const uint64_t round_constants[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  // pi-lane walk with the rho rotation of each step (FIPS 202 sec 3.2.2-3.2.3)
  static const unsigned walk[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  static const unsigned rot[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  for (int r = 0; r < 24; ++r) {
    uint64_t col[5];
    for (int x = 0; x < 5; ++x) col[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
    for (int x = 0; x < 5; ++x) {
      const uint64_t d = col[(x + 4) % 5] ^ rotl64(col[(x + 1) % 5], 1);
      for (int y = 0; y < 25; y += 5) s[y + x] ^= d;
    }
    uint64_t carry = s[1];
    for (int i = 0; i < 24; ++i) {
      const uint64_t next = s[walk[i]];
      s[walk[i]] = rotl64(carry, rot[i]);
      carry = next;
    }
    for (int y = 0; y < 25; y += 5) {
      uint64_t row[5];
      for (int x = 0; x < 5; ++x) row[x] = s[y + x];
      for (int x = 0; x < 5; ++x) s[y + x] = row[x] ^ (~row[(x + 1) % 5] & row[(x + 2) % 5]);
    }
    s[0] ^= round_constants[r];
  }
}

class Sponge {
 public:
  Sponge(unsigned rate, uint8_t suffix) : rate_(rate), suffix_(suffix) { std::memset(s_, 0, sizeof s_); }
  void absorb(const uint8_t* in, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      s_[pos_ >> 3] ^= (uint64_t)in[i] << (8 * (pos_ & 7));
      if (++pos_ == rate_) { keccak_f1600(s_); pos_ = 0; }
    }
  }
  void squeeze(uint8_t* out, size_t n) {
    if (!squeezing_) {
      s_[pos_ >> 3] ^= (uint64_t)suffix_ << (8 * (pos_ & 7));
      s_[(rate_ - 1) >> 3] ^= 0x80ULL << (8 * ((rate_ - 1) & 7));
      keccak_f1600(s_);
      pos_ = 0;
      squeezing_ = true;
    }
    for (size_t i = 0; i < n; ++i) {
      if (pos_ == rate_) { keccak_f1600(s_); pos_ = 0; }
      out[i] = (uint8_t)(s_[pos_ >> 3] >> (8 * (pos_ & 7)));
      ++pos_;
    }
  }

 private:
  uint64_t s_[25];
  unsigned rate_, pos_ = 0;
  uint8_t suffix_;
  bool squeezing_ = false;
};

inline Sponge sha3_512_sponge() { return Sponge(72, 0x06); }
inline Sponge shake256_sponge() { return Sponge(136, 0x1F); }

}  // namespace zk
