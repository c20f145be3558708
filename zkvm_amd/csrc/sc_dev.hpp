// sc_dev.hpp -- integers mod l for the device (and the host tests): eight 32-bit
// limbs in Montgomery form (R = 2^260); the product works on ten 26-bit limbs
// (carry-free v_mad_u64_u32 columns).  This is the verifier's challenge algebra moved onto the GPU
// (curve25519-dalek `Scalar`, RFC 9496 sec 4.4).
#pragma once
#include "field.hpp"   // ZK_HD, ZK_UNROLL

namespace zk {

struct scm {          // value * 2^260 mod l, always < l
  uint32_t v[8];
};

#define ZK_SC_L { 0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0x00000000u, 0x00000000u, 0x00000000u, 0x10000000u }
// Montgomery radix R = 2^260 (ten 26-bit limbs inside the product, see scm_mul_core)
#define ZK_SC_NPRIME26 0x2547e1bu   // -l^-1 mod 2^26
#define ZK_SC_L26 { 0x0f5d3edu, 0x098c697u, 0x1cd6581u, 0x37a8bdeu, 0x014def9u, 0u, 0u, 0u, 0u, 0x0040000u }
#define ZK_SC_R1 { 0x6721e6edu, 0x45af48bdu, 0xab5ac67eu, 0x35e51b3bu, 0xffffffebu, 0xffffffffu, 0xffffffffu, 0x0fffffffu }
#define ZK_SC_R2 { 0xe952d13bu, 0x69f9d265u, 0x3c715beau, 0x687604d6u, 0xf5be65cbu, 0xec73d217u, 0x309a3dceu, 0x09411b7cu }
#define ZK_SC_R3 { 0x5bbc47ffu, 0xd5d6c1e6u, 0x3074a06du, 0xd7af6287u, 0xec5b6514u, 0x65dc6c04u, 0x99cec780u, 0x030b7735u }
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

// Montgomery product a * b / R mod l, R = 2^260.  a < 2^256, b < l  ->  result < l.
// The operands are split into ten 26-bit limbs, so that -- exactly as in fe_mul -- every partial
// product is one v_mad_u64_u32 into a 64-bit column that cannot overflow (<= 16 products of 52 bits),
// with no carries and no register pairing inside the product; the reduction adds m_i * l column by
// column (l = 2^252 + c has only limbs 0..4 and 9 non-zero: six products per step).  The operand-
// scanning form on 32-bit limbs compiled to 629 VALU instructions (96 multiply-adds, 133 64-bit adds,
// 321 moves pairing registers around 32-bit carries).
ZK_HD void scm_limbs26(uint32_t out[10], const uint32_t w[8]) {
  ZK_UNROLL for (int i = 0; i < 10; ++i) {
    const int o = 26 * i, k = o >> 5, sft = o & 31;
    uint64_t two = w[k];
    if (k + 1 < 8) two |= (uint64_t)w[k + 1] << 32;
    out[i] = (uint32_t)(two >> sft) & 0x3ffffffu;
  }
}

ZK_HD scm scm_mul_core(const scm& av, const scm& bv) {
  const uint32_t l26[10] = ZK_SC_L26;
  uint32_t A[10], B[10];
  scm_limbs26(A, av.v);
  scm_limbs26(B, bv.v);
  uint64_t t[20];
  ZK_UNROLL for (int k = 0; k < 20; ++k) t[k] = 0;
  ZK_UNROLL for (int i = 0; i < 10; ++i)
    ZK_UNROLL for (int j = 0; j < 10; ++j) t[i + j] += (uint64_t)A[i] * B[j];
  ZK_UNROLL for (int i = 0; i < 10; ++i) {
    const uint32_t m = ((uint32_t)t[i] * ZK_SC_NPRIME26) & 0x3ffffffu;
    ZK_UNROLL for (int j = 0; j < 10; ++j)
      if (j <= 4 || j == 9) t[i + j] += (uint64_t)m * l26[j];
    t[i + 1] += t[i] >> 26;               // the low 26 bits of t[i] are zero now
  }
  // limbs 10..19 -> eight 32-bit words
  ZK_UNROLL for (int k = 10; k < 19; ++k) { t[k + 1] += t[k] >> 26; t[k] &= 0x3ffffffu; }
  scm r;
  ZK_UNROLL for (int i = 0; i < 8; ++i) {
    // word i = bits [32 i, 32 i + 32) of sum_k t[10 + k] 2^(26 k)
    const int lo_limb = (32 * i) / 26, sft = (32 * i) % 26;
    uint64_t v = t[10 + lo_limb] >> sft;
    v |= t[10 + lo_limb + 1] << (26 - sft);
    if (10 + lo_limb + 2 < 20) v |= t[10 + lo_limb + 2] << (52 - sft);
    r.v[i] = (uint32_t)v;
  }
  // result < 2l < 2^254: fits the eight words
  scm_cond_sub_l(r.v, scm_geq_l(r.v));
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

// ---- lazy form: ten 26-bit limbs, any representative of the residue class ---------------------------------
// scm pays for its canonical range at every step: a product splits both operands into 26-bit limbs, packs the
// result back into eight words and compares / subtracts l (366 VALU instructions, 150 of them multiply-adds), an
// addition is a carry chain, a compare and a conditional subtraction (125).  With R = 2^260 a Montgomery product
// tolerates operands far beyond l:   (A B + m l) / R  <  A B / 2^260 + l,   so values may float in [0, 2^260)
// and stay in limb form between operations: product = the multiply-adds and one carry pass, sum = ten additions,
// difference = a + 256 l - b limb by limb.  Only what leaves the kernel is reduced to [0, l) (scl_canon_words).
//
// Bounds (checked where they are used): "tight" = limbs 0..8 < 2^26, limb 9 < 2^26 - 1 (value < 2^260 - 2^235).
//   scl_mul(a, b)   needs  max limb(a) * max limb(b) <= 2^60  (columns of ten products stay below 2^64) and
//                   value(a) * value(b) < 2^519;  result tight, < a b / 2^260 + l
//   scl_sub(a, b)   needs b tight; limbs grow by < 2^27.6, the value by 2^260
//   scl_weak(a)     any limbs < 2^32 whose carried top limb fits 32 bits (value < 2^265): result tight, < 2 l
struct scl {
  uint32_t v[10];
};

#define ZK_SC_M256L { 0x9d3ed00u, 0x8c6973bu, 0x9658124u, 0xa8bde71u, 0x8def9dcu, 0x8000003u, 0x7fffffeu, 0x7fffffeu, 0x7fffffeu, 0x3fffffeu }
#define ZK_SC_ONE26 { 0x321e6edu, 0x3d22f59u, 0x067e45au, 0x0eead6bu, 0x335e51bu, 0x3fffffau, 0x3ffffffu, 0x3ffffffu, 0x3ffffffu, 0x003ffffu }
#define ZK_SC_R2_26 { 0x152d13bu, 0x274997au, 0x1bea69fu, 0x358f1c5u, 0x3687604u, 0x16f9972u, 0x33d217fu, 0x0f73bb1u, 0x37c309au, 0x0025046u }

ZK_HD scl scl_zero() { scl r; ZK_UNROLL for (int i = 0; i < 10; ++i) r.v[i] = 0; return r; }
ZK_HD scl scl_one() { const scl r = {ZK_SC_ONE26}; return r; }              // 1 in Montgomery form
ZK_HD scl scl_plain_one() { scl r = scl_zero(); r.v[0] = 1; return r; }      // the integer 1: a * it = a / R, i.e. Montgomery -> plain
ZK_HD scl scl_r2() { const scl r = {ZK_SC_R2_26}; return r; }                // plain -> Montgomery
ZK_HD scl scl_from_words(const uint32_t w[8]) { scl r; scm_limbs26(r.v, w); return r; }
ZK_HD scl scl_from_scm(const scm& a) { return scl_from_words(a.v); }

ZK_HD void scl_carry(scl& a) {      // limbs 0..8 -> < 2^26; limb 9 takes what is left
  ZK_UNROLL for (int i = 0; i < 9; ++i) { a.v[i + 1] += a.v[i] >> 26; a.v[i] &= 0x3ffffffu; }
}
ZK_HD scl scl_add(const scl& a, const scl& b) { scl r; ZK_UNROLL for (int i = 0; i < 10; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
ZK_HD scl scl_add_c(const scl& a, const scl& b) { scl r = scl_add(a, b); scl_carry(r); return r; }
ZK_HD scl scl_sub(const scl& a, const scl& b) {
  const uint32_t m[10] = ZK_SC_M256L;
  scl r;
  ZK_UNROLL for (int i = 0; i < 10; ++i) r.v[i] = a.v[i] + (m[i] - b.v[i]);
  return r;
}
ZK_HD scl scl_neg(const scl& b) {
  const uint32_t m[10] = ZK_SC_M256L;
  scl r;
  ZK_UNROLL for (int i = 0; i < 10; ++i) r.v[i] = m[i] - b.v[i];
  return r;
}
ZK_HD scl scl_cneg(const scl& b, bool negate) {
  const uint32_t m[10] = ZK_SC_M256L;
  scl r;
  ZK_UNROLL for (int i = 0; i < 10; ++i) r.v[i] = negate ? m[i] - b.v[i] : b.v[i];
  return r;
}

ZK_HD scl scl_mul(const scl& a, const scl& b) {
  const uint32_t l26[10] = ZK_SC_L26;
  uint64_t t[20];
  ZK_UNROLL for (int k = 0; k < 20; ++k) t[k] = 0;
  ZK_UNROLL for (int i = 0; i < 10; ++i)
    ZK_UNROLL for (int j = 0; j < 10; ++j) t[i + j] += (uint64_t)a.v[i] * b.v[j];
  ZK_UNROLL for (int i = 0; i < 10; ++i) {
    const uint32_t m = ((uint32_t)t[i] * ZK_SC_NPRIME26) & 0x3ffffffu;
    ZK_UNROLL for (int j = 0; j < 10; ++j)
      if (j <= 4 || j == 9) t[i + j] += (uint64_t)m * l26[j];
    t[i + 1] += t[i] >> 26;               // the low 26 bits of t[i] are zero now
  }
  scl r;
  ZK_UNROLL for (int k = 10; k < 19; ++k) { t[k + 1] += t[k] >> 26; r.v[k - 10] = (uint32_t)t[k] & 0x3ffffffu; }
  r.v[9] = (uint32_t)t[19];
  return r;
}
ZK_HD scl scl_sq(const scl& a) { return scl_mul(a, a); }

// low 252 bits + l - q c  with q = value >> 252 (l = 2^252 + c):  the same residue, in (0, 2 l)
ZK_HD scl scl_weak(const scl& x) {
  const uint32_t l26[10] = ZK_SC_L26;
  scl a = x;
  scl_carry(a);
  const uint32_t q = a.v[9] >> 18;        // < 2^14
  a.v[9] &= 0x3ffffu;
  // q c as 26-bit limbs (c = limbs 0..4 of l, 125 bits; q c < 2^139: six limbs)
  uint32_t qc[6];
  uint64_t acc = 0;
  ZK_UNROLL for (int i = 0; i < 5; ++i) { acc += (uint64_t)q * l26[i]; qc[i] = (uint32_t)acc & 0x3ffffffu; acc >>= 26; }
  qc[5] = (uint32_t)acc;
  scl r;
  int32_t carry = 0;
  ZK_UNROLL for (int i = 0; i < 10; ++i) {
    const int32_t t = (int32_t)(a.v[i] + l26[i]) - (int32_t)(i < 6 ? qc[i] : 0u) + carry;   // |t| < 2^28
    if (i < 9) { r.v[i] = (uint32_t)t & 0x3ffffffu; carry = t >> 26; }
    else r.v[i] = (uint32_t)t;            // >= 0: the value is positive
  }
  return r;
}

// the canonical representative as eight little-endian words
ZK_HD void scl_canon_words(uint32_t out[8], const scl& x) {
  const uint32_t l26[10] = ZK_SC_L26;
  const scl r = scl_weak(x);              // in (0, 2 l)
  uint32_t d[10];
  int32_t borrow = 0;
  ZK_UNROLL for (int i = 0; i < 10; ++i) {
    const int32_t t = (int32_t)r.v[i] - (int32_t)l26[i] + borrow;
    if (i < 9) { d[i] = (uint32_t)t & 0x3ffffffu; borrow = t >> 26; }
    else { d[i] = (uint32_t)t; borrow = t >> 31; }
  }
  const bool keep = borrow != 0;          // r < l
  uint32_t f[10];
  ZK_UNROLL for (int i = 0; i < 10; ++i) f[i] = keep ? r.v[i] : d[i];
  ZK_UNROLL for (int i = 0; i < 8; ++i) {
    const int lo = (32 * i) / 26, sft = (32 * i) % 26;
    uint32_t w = f[lo] >> sft;
    w |= f[lo + 1] << (26 - sft);
    if (lo + 2 < 10 && 52 - sft < 32) w |= f[lo + 2] << (52 - sft);
    out[i] = w;
  }
}
// tight limbs of a value < 2^256 <-> eight words (the compact storage of tables in LDS)
ZK_HD void scl_pack8(uint32_t out[8], const scl& f) {
  ZK_UNROLL for (int i = 0; i < 8; ++i) {
    const int lo = (32 * i) / 26, sft = (32 * i) % 26;
    uint32_t w = f.v[lo] >> sft;
    w |= f.v[lo + 1] << (26 - sft);
    if (lo + 2 < 10 && 52 - sft < 32) w |= f.v[lo + 2] << (52 - sft);
    out[i] = w;
  }
}
ZK_HD scm scl_to_scm(const scl& a) { scm r; scl_canon_words(r.v, a); return r; }   // same value, canonical range

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
