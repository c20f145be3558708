/* fe51.c -- GF(2^255-19) in five 51-bit limbs with 128-bit products.
 *
 * TEST INFRASTRUCTURE (see oracle.h).  Restates the field layer the reference
 * path rests on: curve25519-dalek's u64 backend `FieldElement51` (not mounted
 * under /root/reference -- SURVEY.md sec 8(a) rows a1, a2).  Written from RFC
 * 9496 sec 4.1-4.2 and RFC 7748 sec 4.1; checked limb-for-value against
 * oracle/pyref.py (Python ints) in tests/test_oracle_field.py.
 *
 * Invariants: every fe handed between functions has limbs < 2^52 ("reduced
 * weakly"); fe_mul / fe_sq accept limbs < 2^54.
 */
#include "oracle.h"
#include "constants.h"
#include <string.h>

typedef unsigned __int128 u128;
#define M51 ((1ULL << 51) - 1)

static const fe FE_ONE_C = {{1, 0, 0, 0, 0}};
static const fe FE_SQRT_M1_C = FE_SQRT_M1;

static void fe_weak_carry(fe *h) {
  uint64_t c;
  c = h->v[0] >> 51; h->v[0] &= M51; h->v[1] += c;
  c = h->v[1] >> 51; h->v[1] &= M51; h->v[2] += c;
  c = h->v[2] >> 51; h->v[2] &= M51; h->v[3] += c;
  c = h->v[3] >> 51; h->v[3] &= M51; h->v[4] += c;
  c = h->v[4] >> 51; h->v[4] &= M51; h->v[0] += 19 * c;
}

void fe_frombytes(fe *h, const uint8_t s[32]) {
  uint64_t w[4];
  for (int i = 0; i < 4; ++i) {
    w[i] = 0;
    for (int j = 7; j >= 0; --j) w[i] = (w[i] << 8) | s[8 * i + j];
  }
  h->v[0] = w[0] & M51;
  h->v[1] = ((w[0] >> 51) | (w[1] << 13)) & M51;
  h->v[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
  h->v[3] = ((w[2] >> 25) | (w[3] << 39)) & M51;
  h->v[4] = (w[3] >> 12) & M51; /* drops bit 255 */
}

void fe_tobytes(uint8_t s[32], const fe *f) {
  fe h = *f;
  fe_weak_carry(&h);
  fe_weak_carry(&h);
  /* h < 2^255 + small; q = 1 iff h >= p */
  uint64_t q = (h.v[0] + 19) >> 51;
  q = (h.v[1] + q) >> 51;
  q = (h.v[2] + q) >> 51;
  q = (h.v[3] + q) >> 51;
  q = (h.v[4] + q) >> 51;
  h.v[0] += 19 * q;
  uint64_t c;
  c = h.v[0] >> 51; h.v[0] &= M51; h.v[1] += c;
  c = h.v[1] >> 51; h.v[1] &= M51; h.v[2] += c;
  c = h.v[2] >> 51; h.v[2] &= M51; h.v[3] += c;
  c = h.v[3] >> 51; h.v[3] &= M51; h.v[4] += c;
  h.v[4] &= M51; /* discard 2^255 */
  uint64_t w[4];
  w[0] = h.v[0] | (h.v[1] << 51);
  w[1] = (h.v[1] >> 13) | (h.v[2] << 38);
  w[2] = (h.v[2] >> 26) | (h.v[3] << 25);
  w[3] = (h.v[3] >> 39) | (h.v[4] << 12);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) s[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}

void fe_add(fe *h, const fe *f, const fe *g) {
  for (int i = 0; i < 5; ++i) h->v[i] = f->v[i] + g->v[i];
  fe_weak_carry(h);
}

void fe_sub(fe *h, const fe *f, const fe *g) {
  /* add 8p limb-wise so no limb underflows for g limbs < 2^54 */
  h->v[0] = f->v[0] + 8 * (M51 - 18) - g->v[0];
  for (int i = 1; i < 5; ++i) h->v[i] = f->v[i] + 8 * M51 - g->v[i];
  fe_weak_carry(h);
}

void fe_neg(fe *h, const fe *f) {
  fe z = {{0, 0, 0, 0, 0}};
  fe_sub(h, &z, f);
}

void fe_mul(fe *h, const fe *f, const fe *g) {
  u128 t[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 5; ++i) {
    for (int j = 0; j < 5; ++j) {
      int k = i + j;
      if (k < 5) t[k] += (u128)f->v[i] * g->v[j];
      else t[k - 5] += (u128)f->v[i] * (19 * g->v[j]);
    }
  }
  uint64_t r[5];
  u128 c;
  c = t[0] >> 51; r[0] = (uint64_t)t[0] & M51; t[1] += c;
  c = t[1] >> 51; r[1] = (uint64_t)t[1] & M51; t[2] += c;
  c = t[2] >> 51; r[2] = (uint64_t)t[2] & M51; t[3] += c;
  c = t[3] >> 51; r[3] = (uint64_t)t[3] & M51; t[4] += c;
  c = t[4] >> 51; r[4] = (uint64_t)t[4] & M51;
  u128 w = (u128)r[0] + c * 19;
  r[0] = (uint64_t)w & M51;
  r[1] += (uint64_t)(w >> 51);
  for (int i = 0; i < 5; ++i) h->v[i] = r[i];
}

void fe_sq(fe *h, const fe *f) { fe_mul(h, f, f); }

static void fe_sqn(fe *h, const fe *f, int n) {
  fe_sq(h, f);
  for (int i = 1; i < n; ++i) fe_sq(h, h);
}

/* z^(2^250-1) and z^11, the shared prefix of both fixed exponents */
static void fe_pow_2_250_1(fe *out, fe *z11, const fe *z) {
  fe z2, z9, t, z2_5_0, z2_10_0, z2_20_0, z2_50_0, z2_100_0;
  fe_sq(&z2, z);
  fe_sqn(&t, &z2, 2);
  fe_mul(&z9, &t, z);
  fe_mul(z11, &z9, &z2);
  fe_sq(&t, z11);
  fe_mul(&z2_5_0, &t, &z9);
  fe_sqn(&t, &z2_5_0, 5);
  fe_mul(&z2_10_0, &t, &z2_5_0);
  fe_sqn(&t, &z2_10_0, 10);
  fe_mul(&z2_20_0, &t, &z2_10_0);
  fe_sqn(&t, &z2_20_0, 20);
  fe_mul(&t, &t, &z2_20_0);
  fe_sqn(&t, &t, 10);
  fe_mul(&z2_50_0, &t, &z2_10_0);
  fe_sqn(&t, &z2_50_0, 50);
  fe_mul(&z2_100_0, &t, &z2_50_0);
  fe_sqn(&t, &z2_100_0, 100);
  fe_mul(&t, &t, &z2_100_0);
  fe_sqn(&t, &t, 50);
  fe_mul(out, &t, &z2_50_0);
}

void fe_invert(fe *h, const fe *f) {
  fe t, z11;
  fe_pow_2_250_1(&t, &z11, f);
  fe_sqn(&t, &t, 5);
  fe_mul(h, &t, &z11); /* 2^255 - 21 = p - 2 */
}

void fe_pow22523(fe *h, const fe *f) {
  fe t, z11;
  fe_pow_2_250_1(&t, &z11, f);
  fe_sqn(&t, &t, 2);
  fe_mul(h, &t, f); /* 2^252 - 3 = (p-5)/8 */
}

int fe_is_negative(const fe *f) {
  uint8_t s[32];
  fe_tobytes(s, f);
  return s[0] & 1;
}

int fe_is_zero(const fe *f) {
  uint8_t s[32];
  fe_tobytes(s, f);
  uint8_t acc = 0;
  for (int i = 0; i < 32; ++i) acc |= s[i];
  return acc == 0;
}

int fe_eq(const fe *f, const fe *g) {
  uint8_t a[32], b[32];
  fe_tobytes(a, f);
  fe_tobytes(b, g);
  return memcmp(a, b, 32) == 0;
}

/* SQRT_RATIO_M1(u, v), RFC 9496 sec 4.2.  Returns was_square; *r is the
 * non-negative root of u/v (or of SQRT_M1*u/v when u/v is not a square). */
int fe_sqrt_ratio_m1(fe *r, const fe *u, const fe *v) {
  fe v3, v7, t, check, neg_u, neg_u_i, r_prime;
  fe_sq(&t, v);
  fe_mul(&v3, &t, v);
  fe_sq(&t, &v3);
  fe_mul(&v7, &t, v);
  fe_mul(&t, u, &v7);
  fe_pow22523(&t, &t);
  fe_mul(&t, &t, &v3);
  fe_mul(r, &t, u);
  fe_sq(&t, r);
  fe_mul(&check, &t, v);
  fe_neg(&neg_u, u);
  fe_mul(&neg_u_i, &neg_u, &FE_SQRT_M1_C);
  int correct_sign = fe_eq(&check, u);
  int flipped_sign = fe_eq(&check, &neg_u);
  int flipped_sign_i = fe_eq(&check, &neg_u_i);
  fe_mul(&r_prime, r, &FE_SQRT_M1_C);
  if (flipped_sign | flipped_sign_i) *r = r_prime;
  if (fe_is_negative(r)) fe_neg(r, r);
  (void)FE_ONE_C;
  return correct_sign | flipped_sign;
}
