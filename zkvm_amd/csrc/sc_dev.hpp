// sc_dev.hpp -- integers mod l for the device (and the host tests): eight 32-bit
// limbs in Montgomery form (R = 2^256), CIOS multiplication built on
// v_mad_u64_u32.  This is the verifier's challenge algebra moved onto the GPU
// (curve25519-dalek `Scalar`, RFC 9496 sec 4.4).
#pragma once
#include "field.hpp"   // ZK_HD, ZK_UNROLL

namespace zk {

struct scm {          // value * 2^256 mod l, always < l
  uint32_t v[8];
};

#define ZK_SC_L { 0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0x00000000u, 0x00000000u, 0x00000000u, 0x10000000u }
#define ZK_SC_NPRIME 0x12547e1bu
#define ZK_SC_R1 { 0x8d98951du, 0xd6ec3174u, 0x737dcf70u, 0xc6ef5bf4u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0x0fffffffu }
#define ZK_SC_R2 { 0x449c0f01u, 0xa40611e3u, 0x68859347u, 0xd00e1ba7u, 0x17f5be65u, 0xceec73d2u, 0x7c309a3du, 0x0399411bu }
#define ZK_SC_R3 { 0x7b83a2dbu, 0x2a9e4968u, 0xaef7f3ecu, 0x278324e6u, 0x04ec5b65u, 0x8065dc6cu, 0x3599cec7u, 0x0e530b77u }
#define ZK_SC_LM2 { 0x5cf5d3ebu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0x00000000u, 0x00000000u, 0x00000000u, 0x10000000u }

ZK_HD scm scm_zero() { scm r; ZK_UNROLL for (int i = 0; i < 8; ++i) r.v[i] = 0; return r; }
ZK_HD scm scm_one() { const scm r = {ZK_SC_R1}; return r; }

// a >= l ?
ZK_HD bool scm_geq_l(const uint32_t a[8]) {
  const uint32_t l[8] = ZK_SC_L;
  bool ge = true, decided = false;
  ZK_UNROLL for (int i = 7; i >= 0; --i) {
    const bool gt = a[i] > l[i], lt = a[i] < l[i];
    ge = decided ? ge : (gt ? true : (lt ? false : ge));
    decided = decided | gt | lt;
  }
  return ge;
}

ZK_HD void scm_cond_sub_l(uint32_t a[8], bool doit) {
  const uint32_t l[8] = ZK_SC_L;
  uint64_t borrow = 0;
  ZK_UNROLL for (int i = 0; i < 8; ++i) {
    const uint64_t d = (uint64_t)a[i] - (doit ? l[i] : 0u) - borrow;
    a[i] = (uint32_t)d;
    borrow = (d >> 32) & 1;
  }
}

ZK_HD scm scm_add(const scm& a, const scm& b) {
  scm r;
  uint64_t c = 0;
  ZK_UNROLL for (int i = 0; i < 8; ++i) { c += (uint64_t)a.v[i] + b.v[i]; r.v[i] = (uint32_t)c; c >>= 32; }
  scm_cond_sub_l(r.v, scm_geq_l(r.v));   // a + b < 2l < 2^254: no carry out
  return r;
}

ZK_HD scm scm_sub(const scm& a, const scm& b) {
  const uint32_t l[8] = ZK_SC_L;
  scm r;
  uint64_t borrow = 0;
  ZK_UNROLL for (int i = 0; i < 8; ++i) {
    const uint64_t d = (uint64_t)a.v[i] - b.v[i] - borrow;
    r.v[i] = (uint32_t)d;
    borrow = (d >> 32) & 1;
  }
  const bool neg = borrow != 0;
  uint64_t c = 0;
  ZK_UNROLL for (int i = 0; i < 8; ++i) { c += (uint64_t)r.v[i] + (neg ? l[i] : 0u); r.v[i] = (uint32_t)c; c >>= 32; }
  return r;
}

ZK_HD scm scm_neg(const scm& a) { return scm_sub(scm_zero(), a); }

// Montgomery product a * b / 2^256 mod l (CIOS).  a < 2^256, b < l  ->  result < l.
// (629 VALU instructions as compiled, half of them moves that pair registers around the 32-bit
// carries.  Measured alternatives, all slower in k_prepare: a real non-inlined function (0.30 -> 0.38 ms,
// although it shrinks the kernel from 150 KB to 19 KB of code); product scanning with the carry-out of
// v_mad_u64_u32 taken through inline asm (336 instructions, but one dependent multiply-add chain:
// 0.25 -> 0.34 ms; two chains per column: 0.37 ms).  The moves are full-rate and the operand-scanning
// form keeps ~8 independent multiply-adds in flight.)
ZK_HD scm scm_mul_core(const scm& av, const scm& bv) {
  const uint32_t* a = av.v;
  const uint32_t* b = bv.v;
  const uint32_t l[8] = ZK_SC_L;
  uint32_t t[10];
  ZK_UNROLL for (int i = 0; i < 10; ++i) t[i] = 0;
  ZK_UNROLL for (int i = 0; i < 8; ++i) {
    uint64_t c = 0;
    ZK_UNROLL for (int j = 0; j < 8; ++j) {
      c += (uint64_t)a[j] * b[i] + t[j];
      t[j] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[8] = (uint32_t)c;
    t[9] = (uint32_t)(c >> 32);
    const uint32_t m = t[0] * ZK_SC_NPRIME;
    c = (uint64_t)m * l[0] + t[0];
    c >>= 32;
    ZK_UNROLL for (int j = 1; j < 8; ++j) {
      c += (uint64_t)m * l[j] + t[j];
      t[j - 1] = (uint32_t)c;
      c >>= 32;
    }
    c += t[8];
    t[7] = (uint32_t)c;
    t[8] = t[9] + (uint32_t)(c >> 32);
  }
  scm r;
  ZK_UNROLL for (int i = 0; i < 8; ++i) r.v[i] = t[i];
  // t < 2l (t[8] can only be 0 here because 2l < 2^256)
  scm_cond_sub_l(r.v, t[8] != 0 || scm_geq_l(r.v));
  return r;
}

ZK_HD scm scm_mul_raw(const uint32_t a[8], const uint32_t b[8]) {
  scm x, y;
  ZK_UNROLL for (int i = 0; i < 8; ++i) { x.v[i] = a[i]; y.v[i] = b[i]; }
  return scm_mul_core(x, y);
}
ZK_HD scm scm_mul(const scm& a, const scm& b) { return scm_mul_core(a, b); }
ZK_HD scm scm_sq(const scm& a) { return scm_mul_core(a, a); }

// plain little-endian words (any value < 2^256) -> Montgomery form of (value mod l)
ZK_HD scm scm_from_words(const uint32_t w[8]) {
  const uint32_t r2[8] = ZK_SC_R2;
  return scm_mul_raw(w, r2);
}

// canonical check: value < l
ZK_HD bool scm_is_canonical(const uint32_t w[8]) { return !scm_geq_l(w); }

// 64 bytes (16 words) little endian, reduced mod l (Scalar::from_bytes_mod_order_wide)
ZK_HD scm scm_from_wide(const uint32_t w[16]) {
  const uint32_t r2[8] = ZK_SC_R2, r3[8] = ZK_SC_R3;
  const scm lo = scm_mul_raw(w, r2);        // lo * R
  const scm hi = scm_mul_raw(w + 8, r3);    // hi * 2^256 * R
  return scm_add(lo, hi);
}

// Montgomery form -> canonical little-endian words
ZK_HD void scm_to_words(uint32_t out[8], const scm& a) {
  const uint32_t one[8] = {1, 0, 0, 0, 0, 0, 0, 0};
  const scm r = scm_mul_raw(a.v, one);
  ZK_UNROLL for (int i = 0; i < 8; ++i) out[i] = r.v[i];
}

ZK_HD scm scm_from_u32(uint32_t x) {
  const uint32_t w[8] = {x, 0, 0, 0, 0, 0, 0, 0};
  return scm_from_words(w);
}

// a^(l-2)
ZK_HD scm scm_invert(const scm& a) {
  const uint32_t e[8] = ZK_SC_LM2;
  scm acc = scm_one();
  ZK_NOUNROLL for (int i = 252; i >= 0; --i) {
    acc = scm_sq(acc);
    uint32_t word = 0;
    ZK_UNROLL for (int k = 0; k < 8; ++k) word = (k == (i >> 5)) ? e[k] : word;
    if ((word >> (i & 31)) & 1) acc = scm_mul(acc, a);
  }
  return acc;
}

}  // namespace zk
