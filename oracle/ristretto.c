/* ristretto.c -- edwards25519 group law and the ristretto255 encoding.
 *
 * TEST INFRASTRUCTURE (see oracle.h).  Restates curve25519-dalek
 * `EdwardsPoint` / `RistrettoPoint` / `CompressedRistretto` (SURVEY.md sec 8(a)
 * rows a4, a5; not mounted under /root/reference) from RFC 9496 sec 4.3 and
 * Hisil-Wong-Carter-Dawson 2008 (a = -1 extended coordinates).  Pinned against
 * libsodium 1.0.18 through tests/golden/ristretto255.json.
 */
#include "oracle.h"
#include "constants.h"
#include <string.h>

static const fe C_D = FE_D;
static const fe C_D2 = FE_D2;
static const fe C_SQRT_M1 = FE_SQRT_M1;
static const fe C_ONE_MINUS_D_SQ = FE_ONE_MINUS_D_SQ;
static const fe C_D_MINUS_ONE_SQ = FE_D_MINUS_ONE_SQ;
static const fe C_SQRT_AD_MINUS_ONE = FE_SQRT_AD_MINUS_ONE;
static const fe C_INVSQRT_A_MINUS_D = FE_INVSQRT_A_MINUS_D;
static const fe C_ONE = {{1, 0, 0, 0, 0}};
static const fe C_ZERO = {{0, 0, 0, 0, 0}};

void ge_identity(ge *p) { p->X = C_ZERO; p->Y = C_ONE; p->Z = C_ONE; p->T = C_ZERO; }

void ge_basepoint(ge *p) {
  static const fe bx = FE_BASE_X, by = FE_BASE_Y, bt = FE_BASE_T;
  p->X = bx; p->Y = by; p->Z = C_ONE; p->T = bt;
}

/* add-2008-hwcd-3 (a = -1), 8M + 1 mul by 2d */
void ge_add(ge *r, const ge *p, const ge *q) {
  fe a, b, c, d, e, f, g, h, t0, t1;
  fe_sub(&t0, &p->Y, &p->X);
  fe_sub(&t1, &q->Y, &q->X);
  fe_mul(&a, &t0, &t1);
  fe_add(&t0, &p->Y, &p->X);
  fe_add(&t1, &q->Y, &q->X);
  fe_mul(&b, &t0, &t1);
  fe_mul(&c, &p->T, &q->T);
  fe_mul(&c, &c, &C_D2);
  fe_mul(&d, &p->Z, &q->Z);
  fe_add(&d, &d, &d);
  fe_sub(&e, &b, &a);
  fe_sub(&f, &d, &c);
  fe_add(&g, &d, &c);
  fe_add(&h, &b, &a);
  fe_mul(&r->X, &e, &f);
  fe_mul(&r->Y, &g, &h);
  fe_mul(&r->Z, &f, &g);
  fe_mul(&r->T, &e, &h);
}

void ge_neg(ge *r, const ge *p) {
  fe_neg(&r->X, &p->X);
  r->Y = p->Y;
  r->Z = p->Z;
  fe_neg(&r->T, &p->T);
}

void ge_sub(ge *r, const ge *p, const ge *q) {
  ge nq;
  ge_neg(&nq, q);
  ge_add(r, p, &nq);
}

/* dbl-2008-hwcd with a = -1: 4M + 4S */
void ge_double(ge *r, const ge *p) {
  fe A, B, C, E, G, F, H, t;
  fe_sq(&A, &p->X);
  fe_sq(&B, &p->Y);
  fe_sq(&C, &p->Z);
  fe_add(&C, &C, &C);
  fe_add(&t, &p->X, &p->Y);
  fe_sq(&t, &t);
  fe_sub(&E, &t, &A);
  fe_sub(&E, &E, &B);   /* E = (X+Y)^2 - A - B */
  fe_sub(&G, &B, &A);   /* G = D + B, D = -A */
  fe_sub(&F, &G, &C);   /* F = G - C */
  fe_add(&H, &A, &B);
  fe_neg(&H, &H);       /* H = D - B */
  fe_mul(&r->X, &E, &F);
  fe_mul(&r->Y, &G, &H);
  fe_mul(&r->T, &E, &H);
  fe_mul(&r->Z, &F, &G);
}

void ge_scalarmult(ge *r, const sc *k, const ge *p) {
  ge acc;
  ge_identity(&acc);
  for (int i = 255; i >= 0; --i) {
    ge_double(&acc, &acc);
    if ((k->v[i / 64] >> (i % 64)) & 1) ge_add(&acc, &acc, p);
  }
  *r = acc;
}

/* RFC 9496 sec 4.3.3 */
int ge_ristretto_eq(const ge *p, const ge *q) {
  fe a, b;
  fe_mul(&a, &p->X, &q->Y);
  fe_mul(&b, &p->Y, &q->X);
  int e1 = fe_eq(&a, &b);
  fe_mul(&a, &p->Y, &q->Y);
  fe_mul(&b, &p->X, &q->X);
  int e2 = fe_eq(&a, &b);
  return e1 | e2;
}

int ge_is_identity(const ge *p) {
  /* equality with (0,1,1,0): X*1 == Y*0  or  Y*1 == X*0 */
  return fe_is_zero(&p->X) | fe_is_zero(&p->Y);
}

/* DECODE, RFC 9496 sec 4.3.1 */
int ristretto_decode(ge *p, const uint8_t sbytes[32]) {
  fe s, ss, u1, u2, u2_sqr, v, t, invsqrt, den_x, den_y, x, y;
  uint8_t chk[32];
  fe_frombytes(&s, sbytes);
  fe_tobytes(chk, &s);
  if (memcmp(chk, sbytes, 32) != 0) return 0; /* non-canonical (incl. bit 255) */
  if (sbytes[0] & 1) return 0;                 /* negative */
  fe_sq(&ss, &s);
  fe_sub(&u1, &C_ONE, &ss);
  fe_add(&u2, &C_ONE, &ss);
  fe_sq(&u2_sqr, &u2);
  fe_sq(&t, &u1);
  fe_mul(&t, &t, &C_D);
  fe_neg(&t, &t);
  fe_sub(&v, &t, &u2_sqr);
  fe_mul(&t, &v, &u2_sqr);
  int was_square = fe_sqrt_ratio_m1(&invsqrt, &C_ONE, &t);
  fe_mul(&den_x, &invsqrt, &u2);
  fe_mul(&den_y, &invsqrt, &den_x);
  fe_mul(&den_y, &den_y, &v);
  fe_mul(&x, &s, &den_x);
  fe_add(&x, &x, &x);
  if (fe_is_negative(&x)) fe_neg(&x, &x);
  fe_mul(&y, &u1, &den_y);
  fe_mul(&t, &x, &y);
  if (!was_square || fe_is_negative(&t) || fe_is_zero(&y)) return 0;
  p->X = x; p->Y = y; p->Z = C_ONE; p->T = t;
  return 1;
}

/* ENCODE, RFC 9496 sec 4.3.2 */
void ristretto_encode(uint8_t out[32], const ge *p) {
  fe u1, u2, t, invsqrt, den1, den2, z_inv, ix0, iy0, enchanted, x, y, den_inv, s;
  fe_add(&u1, &p->Z, &p->Y);
  fe_sub(&t, &p->Z, &p->Y);
  fe_mul(&u1, &u1, &t);
  fe_mul(&u2, &p->X, &p->Y);
  fe_sq(&t, &u2);
  fe_mul(&t, &t, &u1);
  (void)fe_sqrt_ratio_m1(&invsqrt, &C_ONE, &t);
  fe_mul(&den1, &invsqrt, &u1);
  fe_mul(&den2, &invsqrt, &u2);
  fe_mul(&z_inv, &den1, &den2);
  fe_mul(&z_inv, &z_inv, &p->T);
  fe_mul(&ix0, &p->X, &C_SQRT_M1);
  fe_mul(&iy0, &p->Y, &C_SQRT_M1);
  fe_mul(&enchanted, &den1, &C_INVSQRT_A_MINUS_D);
  fe_mul(&t, &p->T, &z_inv);
  if (fe_is_negative(&t)) { x = iy0; y = ix0; den_inv = enchanted; }
  else { x = p->X; y = p->Y; den_inv = den2; }
  fe_mul(&t, &x, &z_inv);
  if (fe_is_negative(&t)) fe_neg(&y, &y);
  fe_sub(&t, &p->Z, &y);
  fe_mul(&s, &den_inv, &t);
  if (fe_is_negative(&s)) fe_neg(&s, &s);
  fe_tobytes(out, &s);
}

/* MAP, RFC 9496 sec 4.3.4 */
static void elligator_map(ge *p, const fe *t0) {
  fe r, u, v, s, s_prime, c, n, w0, w1, w2, w3, t, ss;
  fe_sq(&r, t0);
  fe_mul(&r, &r, &C_SQRT_M1);
  fe_add(&u, &r, &C_ONE);
  fe_mul(&u, &u, &C_ONE_MINUS_D_SQ);
  fe_mul(&t, &r, &C_D);
  fe_add(&t, &t, &C_ONE);
  fe_neg(&t, &t);              /* -1 - r*d */
  fe_add(&v, &r, &C_D);
  fe_mul(&v, &t, &v);
  int was_square = fe_sqrt_ratio_m1(&s, &u, &v);
  fe_mul(&s_prime, &s, t0);
  if (!fe_is_negative(&s_prime)) fe_neg(&s_prime, &s_prime); /* -|s*t| */
  if (!was_square) { s = s_prime; c = r; }
  else fe_neg(&c, &C_ONE);
  fe_sub(&t, &r, &C_ONE);
  fe_mul(&n, &c, &t);
  fe_mul(&n, &n, &C_D_MINUS_ONE_SQ);
  fe_sub(&n, &n, &v);
  fe_mul(&w0, &s, &v);
  fe_add(&w0, &w0, &w0);
  fe_mul(&w1, &n, &C_SQRT_AD_MINUS_ONE);
  fe_sq(&ss, &s);
  fe_sub(&w2, &C_ONE, &ss);
  fe_add(&w3, &C_ONE, &ss);
  fe_mul(&p->X, &w0, &w3);
  fe_mul(&p->Y, &w2, &w1);
  fe_mul(&p->Z, &w1, &w3);
  fe_mul(&p->T, &w0, &w2);
}

void ristretto_from_uniform_bytes(ge *p, const uint8_t b[64]) {
  fe t1, t2;
  ge p1, p2;
  fe_frombytes(&t1, b);        /* masks bit 255, value taken mod p */
  fe_frombytes(&t2, b + 32);
  elligator_map(&p1, &t1);
  elligator_map(&p2, &t2);
  ge_add(p, &p1, &p2);
}
