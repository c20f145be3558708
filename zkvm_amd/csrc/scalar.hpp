// scalar.hpp -- integers mod l = 2^252 + 27742317777372353535851937790883648493,
// host side.  Replaces curve25519-dalek `Scalar` for the verifier's challenge
// algebra (RFC 9496 sec 4.4).  Four 64-bit limbs,
// schoolbook product, Barrett reduction with mu = floor(2^512 / l).
#pragma once
#include <cstdint>
#include <cstring>

namespace zk {

struct Scalar {
  uint64_t v[4];

  static Scalar zero() { return Scalar{{0, 0, 0, 0}}; }
  static Scalar one() { return Scalar{{1, 0, 0, 0}}; }
  static Scalar from_u64(uint64_t x) { return Scalar{{x, 0, 0, 0}}; }

  static const uint64_t* L() {
    static const uint64_t l[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL};
    return l;
  }

  bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
  bool operator==(const Scalar& o) const { return std::memcmp(v, o.v, 32) == 0; }

  void to_bytes(uint8_t out[32]) const {
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 8; ++j) out[8 * i + j] = (uint8_t)(v[i] >> (8 * j));
  }

  // canonical encoding only (dalek Scalar::from_canonical_bytes)
  static bool from_canonical(const uint8_t b[32], Scalar& out) {
    for (int i = 0; i < 4; ++i) {
      out.v[i] = 0;
      for (int j = 7; j >= 0; --j) out.v[i] = (out.v[i] << 8) | b[8 * i + j];
    }
    return !geq_l(out.v);
  }

  // 64 little-endian bytes reduced mod l (Scalar::from_bytes_mod_order_wide)
  static Scalar from_wide(const uint8_t b[64]) {
    uint64_t x[8];
    for (int i = 0; i < 8; ++i) {
      x[i] = 0;
      for (int j = 7; j >= 0; --j) x[i] = (x[i] << 8) | b[8 * i + j];
    }
    return barrett(x);
  }

  friend Scalar operator+(const Scalar& a, const Scalar& b) {
    Scalar r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (unsigned __int128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
    if (geq_l(r.v)) sub_l(r.v);
    return r;
  }
  friend Scalar operator-(const Scalar& a, const Scalar& b) {
    Scalar r;
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 d = (unsigned __int128)a.v[i] - b.v[i] - borrow;
      r.v[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) {
      unsigned __int128 c = 0;
      for (int i = 0; i < 4; ++i) { c += (unsigned __int128)r.v[i] + L()[i]; r.v[i] = (uint64_t)c; c >>= 64; }
    }
    return r;
  }
  Scalar operator-() const { return zero() - *this; }
  friend Scalar operator*(const Scalar& a, const Scalar& b) {
    uint64_t x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 c = 0;
      for (int j = 0; j < 4; ++j) {
        c += (unsigned __int128)a.v[i] * b.v[j] + x[i + j];
        x[i + j] = (uint64_t)c;
        c >>= 64;
      }
      x[i + 4] = (uint64_t)c;
    }
    return barrett(x);
  }
  Scalar& operator+=(const Scalar& o) { *this = *this + o; return *this; }
  Scalar& operator-=(const Scalar& o) { *this = *this - o; return *this; }
  Scalar& operator*=(const Scalar& o) { *this = *this * o; return *this; }

  // a^(l-2)
  Scalar invert() const {
    static const uint64_t e[4] = {0x5812631a5cf5d3ebULL, 0x14def9dea2f79cd6ULL, 0x0000000000000000ULL, 0x1000000000000000ULL};
    Scalar acc = one();
    for (int i = 252; i >= 0; --i) {
      acc = acc * acc;
      if ((e[i >> 6] >> (i & 63)) & 1) acc = acc * *this;
    }
    return acc;
  }

 private:
  static bool geq_l(const uint64_t a[4]) {
    for (int i = 3; i >= 0; --i) {
      if (a[i] > L()[i]) return true;
      if (a[i] < L()[i]) return false;
    }
    return true;
  }
  static void sub_l(uint64_t a[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; ++i) {
      unsigned __int128 d = (unsigned __int128)a[i] - L()[i] - borrow;
      a[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
  }
  // Barrett (HAC 14.42) with b = 2^64, k = 4: x < b^8
  static Scalar barrett(const uint64_t x[8]) {
    static const uint64_t MU[5] = {0xed9ce5a30a2c131bULL, 0x2106215d086329a7ULL, 0xffffffffffffffebULL,
                                   0xffffffffffffffffULL, 0x000000000000000fULL};
    // q1 = floor(x / b^3): limbs 3..7 ; q2 = q1 * mu ; q3 = floor(q2 / b^5)
    uint64_t q2[10] = {0};
    for (int i = 0; i < 5; ++i) {
      unsigned __int128 c = 0;
      for (int j = 0; j < 5; ++j) {
        c += (unsigned __int128)x[3 + i] * MU[j] + q2[i + j];
        q2[i + j] = (uint64_t)c;
        c >>= 64;
      }
      q2[i + 5] = (uint64_t)c;
    }
    const uint64_t* q3 = q2 + 5;  // 5 limbs
    // r2 = q3 * l mod b^5
    uint64_t r2[5] = {0};
    for (int i = 0; i < 5; ++i) {
      unsigned __int128 c = 0;
      for (int j = 0; i + j < 5; ++j) {   // l has four limbs; j = 4 only carries
        c += (unsigned __int128)q3[i] * (j < 4 ? L()[j] : 0) + r2[i + j];
        r2[i + j] = (uint64_t)c;
        c >>= 64;
      }
    }
    // r = (x mod b^5) - r2  (mod b^5)
    uint64_t r[5];
    uint64_t borrow = 0;
    for (int i = 0; i < 5; ++i) {
      unsigned __int128 d = (unsigned __int128)x[i] - r2[i] - borrow;
      r[i] = (uint64_t)d;
      borrow = (uint64_t)(d >> 64) & 1;
    }
    // at most two subtractions of l
    for (int it = 0; it < 3; ++it) {
      bool ge = r[4] != 0 || geq_l(r);
      if (!ge) break;
      uint64_t bw = 0;
      for (int i = 0; i < 5; ++i) {
        unsigned __int128 d = (unsigned __int128)r[i] - (i < 4 ? L()[i] : 0) - bw;
        r[i] = (uint64_t)d;
        bw = (uint64_t)(d >> 64) & 1;
      }
    }
    Scalar out;
    std::memcpy(out.v, r, 32);
    return out;
  }
};

}  // namespace zk
