// field.hpp -- GF(2^255-19) for CDNA4 (gfx950) and for the host-side finish.
//
// Replaces, on the device, curve25519-dalek's `FieldElement51` (SURVEY.md sec
// 8(a) rows a1/a2; the Rust source is not mounted under /root/reference, the
// arithmetic follows RFC 9496 sec 4.1-4.2 / RFC 7748 sec 4.1).
//
// Representation: the reference's five 51-bit limbs, each limb held as a
// (26-bit, 25-bit) pair of 32-bit VGPRs:
//     x = sum_{i<5} (v[2i] + 2^26 v[2i+1]) * 2^(51 i)
// gfx950 has no 64x64->128 multiply; its widest integer multiplier is
// v_mad_u64_u32 (32x32+64->64), measured at the same issue rate as v_fma_f64
// (tools/ubench/valu_rates.hip, profiles/r01_valu_rates.txt).  Splitting each
// 51-bit limb in two makes every partial product of the 5x5 limb schoolbook
// exactly one v_mad_u64_u32 with the accumulate for free: 100 mads per
// multiplication, 55 per squaring, no carries inside the product.
//
// Limb bounds ("tight"): even limbs < 2^26 + 2^18, odd limbs < 2^25 + 2^18.
// fe_mul / fe_sq accept "loose" inputs: even < 2^27.7, odd < 2^26.7, i.e. one
// fe_add or fe_sub of tight values.  Everything returns tight values except
// fe_add / fe_sub (loose) -- use fe_carry() when chaining them.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_HD_NOINLINE __host__ __device__ inline __attribute__((noinline))
#define ZK_UNROLL _Pragma("unroll")
#define ZK_NOUNROLL _Pragma("clang loop unroll(disable)")
#else
#define ZK_HD inline
#define ZK_HD_NOINLINE inline
#define ZK_UNROLL
#define ZK_NOUNROLL
#endif

namespace zk {

struct fe {
  uint32_t v[10];
};

constexpr uint32_t M26 = (1u << 26) - 1;
constexpr uint32_t M25 = (1u << 25) - 1;

ZK_HD fe fe_zero() {
  fe r;
  ZK_UNROLL for (int i = 0; i < 10; ++i) r.v[i] = 0;
  return r;
}
ZK_HD fe fe_one() {
  fe r = fe_zero();
  r.v[0] = 1;
  return r;
}

// one pass of carry propagation on 32-bit limbs; loose -> tight
ZK_HD void fe_carry(fe& h) {
  uint32_t c;
  ZK_UNROLL for (int i = 0; i < 9; ++i) {
    if (i & 1) { c = h.v[i] >> 25; h.v[i] &= M25; }
    else       { c = h.v[i] >> 26; h.v[i] &= M26; }
    h.v[i + 1] += c;
  }
  c = h.v[9] >> 25; h.v[9] &= M25;
  h.v[0] += 19 * c;
}

// h = f + g (no carry: loose result for tight inputs)
ZK_HD void fe_add(fe& h, const fe& f, const fe& g) {
  ZK_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = f.v[i] + g.v[i];
}

// h = f - g + 2p (no carry).  g must be tight; result loose when f is tight.
ZK_HD void fe_sub(fe& h, const fe& f, const fe& g) {
  h.v[0] = f.v[0] + 2 * (M26 - 18) - g.v[0];
  ZK_UNROLL for (int i = 1; i < 10; ++i) h.v[i] = f.v[i] + 2 * ((i & 1) ? M25 : M26) - g.v[i];
}

// h = f - g + 4p, carried.  g may be loose.
ZK_HD void fe_sub_c(fe& h, const fe& f, const fe& g) {
  h.v[0] = f.v[0] + 4 * (M26 - 18) - g.v[0];
  ZK_UNROLL for (int i = 1; i < 10; ++i) h.v[i] = f.v[i] + 4 * ((i & 1) ? M25 : M26) - g.v[i];
  fe_carry(h);
}

// h = f - g + 4p, NOT carried: for a value that is only ever the FIRST operand of a product.  f tight, g < 2^27 (a sum of two
// tight values): limbs < 2^26 + 2^28 (even) / 2^25 + 2^27 (odd), never negative.
ZK_HD void fe_sub4_loose(fe& h, const fe& f, const fe& g) {
  h.v[0] = f.v[0] + 4 * (M26 - 18) - g.v[0];
  ZK_UNROLL for (int i = 1; i < 10; ++i) h.v[i] = f.v[i] + 4 * ((i & 1) ? M25 : M26) - g.v[i];
}

ZK_HD void fe_add_c(fe& h, const fe& f, const fe& g) {
  fe_add(h, f, g);
  fe_carry(h);
}

ZK_HD void fe_neg(fe& h, const fe& f) {  // tight in, tight out
  fe z = fe_zero();
  fe_sub(h, z, f);
  fe_carry(h);
}

// 64-bit column sums -> tight 32-bit limbs
ZK_HD void fe_reduce_cols(fe& h, uint64_t t[10]) {
  uint64_t c;
  ZK_UNROLL for (int i = 0; i < 9; ++i) {
    if (i & 1) { c = t[i] >> 25; h.v[i] = (uint32_t)t[i] & M25; }
    else       { c = t[i] >> 26; h.v[i] = (uint32_t)t[i] & M26; }
    t[i + 1] += c;
  }
  c = t[9] >> 25; h.v[9] = (uint32_t)t[9] & M25;
  uint64_t w = (uint64_t)h.v[0] + 19 * c;   // c < 2^39
  h.v[0] = (uint32_t)w & M26;
  h.v[1] += (uint32_t)(w >> 26);
}

// h = f * g.  100 x (32x32+64->64).
ZK_HD void fe_mul(fe& h, const fe& f, const fe& g) {
  uint32_t g19[10], f2[10];
  ZK_UNROLL for (int j = 0; j < 10; ++j) g19[j] = 19u * g.v[j];
  ZK_UNROLL for (int i = 0; i < 10; ++i) f2[i] = 2u * f.v[i];
  uint64_t t[10];
  ZK_UNROLL for (int k = 0; k < 10; ++k) t[k] = 0;
  ZK_UNROLL for (int i = 0; i < 10; ++i) {
    ZK_UNROLL for (int j = 0; j < 10; ++j) {
      const int k = i + j;
      const uint32_t a = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
      const uint32_t b = (k >= 10) ? g19[j] : g.v[j];
      t[k >= 10 ? k - 10 : k] += (uint64_t)a * b;
    }
  }
  fe_reduce_cols(h, t);
}

// h = f^2.  55 x (32x32+64->64).
ZK_HD void fe_sq(fe& h, const fe& f) {
  uint32_t f2[10], f19[10], f38[10];
  ZK_UNROLL for (int i = 0; i < 10; ++i) { f2[i] = 2u * f.v[i]; f19[i] = 19u * f.v[i]; f38[i] = 38u * f.v[i]; }
  uint64_t t[10];
  ZK_UNROLL for (int k = 0; k < 10; ++k) t[k] = 0;
  ZK_UNROLL for (int i = 0; i < 10; ++i) {
    // diagonal term f_i^2 : weight doubles when i is odd, wraps (x19) when 2i >= 10
    {
      const int k = 2 * i;
      const uint32_t a = (i & 1) ? f2[i] : f.v[i];
      const uint32_t b = (k >= 10) ? f19[i] : f.v[i];
      t[k >= 10 ? k - 10 : k] += (uint64_t)a * b;
    }
    ZK_UNROLL for (int j = i + 1; j < 10; ++j) {
      // 2 f_i f_j, another factor 2 when both odd, x19 on wrap
      const int k = i + j;
      const bool both_odd = (i & 1) && (j & 1);
      // coefficient 2 (x2 when both odd, x19 on wrap), split so that neither
      // 32-bit operand overflows for loose inputs
      uint32_t a, b;
      if (k >= 10) { a = f2[i]; b = both_odd ? f38[j] : f19[j]; }
      else         { a = both_odd ? f2[i] : f.v[i]; b = f2[j]; }
      t[k >= 10 ? k - 10 : k] += (uint64_t)a * b;
    }
  }
  fe_reduce_cols(h, t);
}

ZK_HD void fe_sqn(fe& h, const fe& f, int n) {
  fe_sq(h, f);
  ZK_NOUNROLL for (int i = 1; i < n; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
    // keep every iteration a self-contained squaring: without this fence the optimiser
    // rotates / interleaves the chain and the register allocation triples
    ZK_UNROLL for (int k = 0; k < 10; ++k) asm volatile("" : "+v"(h.v[k]));
#endif
    fe_sq(h, h);
  }
}

// fully reduced limbs (the unique representative in [0, p))
ZK_HD void fe_canon(fe& h, const fe& f) {
  h = f;
  fe_carry(h);
  fe_carry(h);
  // q = 1 iff h >= p  (h < 2^255 + 19*small here)
  uint32_t q = (h.v[0] + 19) >> 26;
  ZK_UNROLL for (int i = 1; i < 10; ++i) q = (h.v[i] + q) >> ((i & 1) ? 25 : 26);
  h.v[0] += 19 * q;
  uint32_t c;
  ZK_UNROLL for (int i = 0; i < 9; ++i) {
    if (i & 1) { c = h.v[i] >> 25; h.v[i] &= M25; }
    else       { c = h.v[i] >> 26; h.v[i] &= M26; }
    h.v[i + 1] += c;
  }
  h.v[9] &= M25;  // drop 2^255
}

ZK_HD bool fe_is_negative(const fe& f) {
  fe c;
  fe_canon(c, f);
  return c.v[0] & 1;
}

ZK_HD bool fe_is_zero(const fe& f) {
  fe c;
  fe_canon(c, f);
  uint32_t acc = 0;
  ZK_UNROLL for (int i = 0; i < 10; ++i) acc |= c.v[i];
  return acc == 0;
}

ZK_HD bool fe_eq(const fe& f, const fe& g) {
  fe d;
  fe_sub_c(d, f, g);
  return fe_is_zero(d);
}

ZK_HD void fe_cswap(fe& a, fe& b, bool swap) {
  ZK_UNROLL for (int i = 0; i < 10; ++i) {
    uint32_t x = a.v[i], y = b.v[i];
    a.v[i] = swap ? y : x;
    b.v[i] = swap ? x : y;
  }
}

ZK_HD void fe_cmov(fe& a, const fe& b, bool mv) {
  ZK_UNROLL for (int i = 0; i < 10; ++i) a.v[i] = mv ? b.v[i] : a.v[i];
}

// 32 little-endian bytes (as 8 x u32 words) -> limbs; bit 255 ignored
ZK_HD void fe_from_words(fe& h, const uint32_t w[8]) {
  // limb k starts at bit ceil(25.5 k)
  const int start[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
  ZK_UNROLL for (int k = 0; k < 10; ++k) {
    const int s = start[k], wi = s >> 5, sh = s & 31;
    uint64_t two = (uint64_t)w[wi] | ((wi + 1 < 8) ? ((uint64_t)w[wi + 1] << 32) : 0);
    h.v[k] = (uint32_t)(two >> sh) & ((k & 1) ? M25 : M26);
  }
}

// canonical limbs -> 8 x u32 little-endian words
ZK_HD void fe_to_words(uint32_t w[8], const fe& f) {
  fe c;
  fe_canon(c, f);
  const int start[10] = {0, 26, 51, 77, 102, 128, 153, 179, 204, 230};
  uint64_t acc[4] = {0, 0, 0, 0};
  ZK_UNROLL for (int k = 0; k < 10; ++k) {
    const int s = start[k], qi = s >> 6, sh = s & 63;
    acc[qi] |= (uint64_t)c.v[k] << sh;
    if (sh > 64 - 26 && qi + 1 < 4) acc[qi + 1] |= (uint64_t)c.v[k] >> (64 - sh);
  }
  ZK_UNROLL for (int i = 0; i < 4; ++i) { w[2 * i] = (uint32_t)acc[i]; w[2 * i + 1] = (uint32_t)(acc[i] >> 32); }
}

// z^(2^250-1) and z^11 (shared prefix of the two fixed exponents)
ZK_HD void fe_pow_2_250_1(fe& out, fe& z11, const fe& z) {
  fe z2, z9, t, a, b, c;
  fe_sq(z2, z);
  fe_sqn(t, z2, 2);
  fe_mul(z9, t, z);
  fe_mul(z11, z9, z2);
  fe_sq(t, z11);
  fe_mul(a, t, z9);        // 2^5 - 1
  fe_sqn(t, a, 5);
  fe_mul(b, t, a);         // 2^10 - 1
  fe_sqn(t, b, 10);
  fe_mul(c, t, b);         // 2^20 - 1
  fe_sqn(t, c, 20);
  fe_mul(t, t, c);         // 2^40 - 1
  fe_sqn(t, t, 10);
  fe_mul(a, t, b);         // 2^50 - 1
  fe_sqn(t, a, 50);
  fe_mul(c, t, a);         // 2^100 - 1
  fe_sqn(t, c, 100);
  fe_mul(t, t, c);         // 2^200 - 1
  fe_sqn(t, t, 50);
  fe_mul(out, t, a);       // 2^250 - 1
}

ZK_HD void fe_invert(fe& h, const fe& f) {
  fe t, z11;
  fe_pow_2_250_1(t, z11, f);
  fe_sqn(t, t, 5);
  fe_mul(h, t, z11);
}

ZK_HD void fe_pow22523(fe& h, const fe& f) {
  fe t, z11;
  fe_pow_2_250_1(t, z11, f);
  fe_sqn(t, t, 2);
  fe_mul(h, t, f);
}

}  // namespace zk
