/* sc25519.c -- integers modulo l = 2^252 + 27742317777372353535851937790883648493.
 *
 * TEST INFRASTRUCTURE (see oracle.h).  Restates curve25519-dalek `Scalar`
 * (not mounted under /root/reference) from RFC 9496
 * sec 4.4.  Plain 4 x 64-bit limbs, schoolbook product, reduction by folding
 * 2^252 = -c (mod l); checked against Python ints and libsodium's scalar vectors in
 * tests/test_oracle_field.py.
 */
#include "oracle.h"
#include <string.h>

typedef unsigned __int128 u128;

/* l and c = l - 2^252, little-endian 64-bit limbs */
static const uint64_t L_[4] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL, 0, 0x1000000000000000ULL};
static const uint64_t C_[2] = {0x5812631a5cf5d3edULL, 0x14def9dea2f79cd6ULL};

/* r[0..n) = a[0..n) + b[0..n), returns carry */
static uint64_t bn_add(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
  u128 c = 0;
  for (int i = 0; i < n; ++i) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t bn_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
  uint64_t borrow = 0;
  for (int i = 0; i < n; ++i) {
    u128 d = (u128)a[i] - b[i] - borrow;
    r[i] = (uint64_t)d;
    borrow = (uint64_t)(d >> 64) & 1;
  }
  return borrow;
}
static int bn_geq(const uint64_t *a, const uint64_t *b, int n) {
  for (int i = n - 1; i >= 0; --i) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
/* r[0..na+nb) = a * b */
static void bn_mul(uint64_t *r, const uint64_t *a, int na, const uint64_t *b, int nb) {
  memset(r, 0, sizeof(uint64_t) * (size_t)(na + nb));
  for (int i = 0; i < na; ++i) {
    u128 c = 0;
    for (int j = 0; j < nb; ++j) {
      c += (u128)a[i] * b[j] + r[i + j];
      r[i + j] = (uint64_t)c;
      c >>= 64;
    }
    r[i + nb] = (uint64_t)c;
  }
}
/* split x (n limbs) at bit 252: lo gets 4 limbs, hi gets n-3 limbs */
static void split252(uint64_t *lo, uint64_t *hi, const uint64_t *x, int n) {
  lo[0] = x[0]; lo[1] = x[1]; lo[2] = x[2]; lo[3] = x[3] & 0x0fffffffffffffffULL;
  for (int i = 3; i < n; ++i) {
    uint64_t next = (i + 1 < n) ? x[i + 1] : 0;
    hi[i - 3] = (x[i] >> 60) | (next << 4);
  }
}

/* x < 2^512 -> x mod l.  With x = x1*2^252 + x0 and 2^252 = -c:
 *   x = x0 - z0 + w0 - w1*c   where z = x1*c = z1*2^252 + z0, w = z1*c = w1*2^252 + w0. */
static void reduce512(uint64_t r[4], const uint64_t x[8]) {
  uint64_t x0[4], x1[5], z[7], z0[4], z1[4], w[6], w0[4], w1[3], v[5], acc[4], t[4];
  split252(x0, x1, x, 8);           /* x1 < 2^260: 5 limbs */
  bn_mul(z, x1, 5, C_, 2);          /* < 2^385: 7 limbs */
  split252(z0, z1, z, 7);           /* z1 < 2^133: limbs 0..2 used (z1[3] = 0) */
  bn_mul(w, z1, 4, C_, 2);          /* < 2^258: 6 limbs, top ones zero */
  split252(w0, w1, w, 6);           /* w1 < 2^6 */
  bn_mul(v, w1, 3, C_, 2);          /* < 2^131 */
  /* acc = x0 + w0 + 2l - z0 - v  (positive, < 2^255) */
  bn_add(acc, x0, w0, 4);
  bn_add(acc, acc, L_, 4);
  bn_add(acc, acc, L_, 4);
  bn_sub(acc, acc, z0, 4);
  t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
  bn_sub(acc, acc, t, 4);
  while (bn_geq(acc, L_, 4)) bn_sub(acc, acc, L_, 4);
  memcpy(r, acc, 32);
}

static void load_le(uint64_t *w, const uint8_t *b, int nwords) {
  for (int i = 0; i < nwords; ++i) {
    w[i] = 0;
    for (int j = 7; j >= 0; --j) w[i] = (w[i] << 8) | b[8 * i + j];
  }
}

void sc_from_bytes_wide(sc *r, const uint8_t b[64]) {
  uint64_t x[8];
  load_le(x, b, 8);
  reduce512(r->v, x);
}

void sc_from_bytes_mod_order(sc *r, const uint8_t b[32]) {
  uint64_t x[8] = {0};
  load_le(x, b, 4);
  reduce512(r->v, x);
}

int sc_from_canonical_bytes(sc *r, const uint8_t b[32]) {
  load_le(r->v, b, 4);
  return !bn_geq(r->v, L_, 4);
}

void sc_to_bytes(uint8_t b[32], const sc *a) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(a->v[i] >> (8 * j));
}

void sc_from_u64(sc *r, uint64_t x) { r->v[0] = x; r->v[1] = r->v[2] = r->v[3] = 0; }

void sc_add(sc *r, const sc *a, const sc *b) {
  uint64_t t[4];
  bn_add(t, a->v, b->v, 4); /* < 2^254, no carry */
  if (bn_geq(t, L_, 4)) bn_sub(t, t, L_, 4);
  memcpy(r->v, t, 32);
}

void sc_sub(sc *r, const sc *a, const sc *b) {
  uint64_t t[4];
  if (bn_sub(t, a->v, b->v, 4)) bn_add(t, t, L_, 4);
  memcpy(r->v, t, 32);
}

void sc_neg(sc *r, const sc *a) {
  sc z = {{0, 0, 0, 0}};
  sc_sub(r, &z, a);
}

void sc_mul(sc *r, const sc *a, const sc *b) {
  uint64_t x[8];
  bn_mul(x, a->v, 4, b->v, 4);
  reduce512(r->v, x);
}

int sc_is_zero(const sc *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
int sc_eq(const sc *a, const sc *b) { return memcmp(a->v, b->v, 32) == 0; }

/* a^(l-2) by left-to-right square and multiply (Fermat) */
void sc_invert(sc *r, const sc *a) {
  uint64_t e[4] = {L_[0] - 2, L_[1], L_[2], L_[3]};
  sc acc;
  sc_from_u64(&acc, 1);
  for (int i = 252; i >= 0; --i) {
    sc_mul(&acc, &acc, &acc);
    if ((e[i / 64] >> (i % 64)) & 1) sc_mul(&acc, &acc, a);
  }
  *r = acc;
}
