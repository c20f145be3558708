// curve.hpp -- edwards25519 group law (a = -1, extended coordinates) and the
// ristretto255 encoding for CDNA4 and the host-side finish.
//
// Replaces curve25519-dalek `EdwardsPoint` / `AffineNielsPoint` /
// `CompressedRistretto::decompress` / `RistrettoPoint::compress` on the
// verification path (Rust source not mounted).
// Formulas: Hisil-Wong-Carter-Dawson 2008 sec 3.1 (add-2008-hwcd-3, dbl-2008-hwcd,
// madd-2008-hwcd-3); encoding: RFC 9496 sec 4.3.1-4.3.4.
#pragma once
#include "field.hpp"
#include "constants.inc"

namespace zk {

struct ge {        // extended (X:Y:Z:T), x = X/Z, y = Y/Z, T = XY/Z
  fe X, Y, Z, T;
};
struct ge_niels {  // affine point prepared for mixed addition: (y+x, y-x, 2dxy)
  fe ypx, ymx, xy2d;
};

#define ZK_CONST_FE(name, init) \
  ZK_HD fe name() { const fe c = {init}; return c; }
ZK_CONST_FE(fe_D, ZK_FE_D)
ZK_CONST_FE(fe_D2, ZK_FE_D2)
ZK_CONST_FE(fe_SQRT_M1, ZK_FE_SQRT_M1)
ZK_CONST_FE(fe_INVSQRT_A_MINUS_D, ZK_FE_INVSQRT_A_MINUS_D)
ZK_CONST_FE(fe_ONE_MINUS_D_SQ, ZK_FE_ONE_MINUS_D_SQ)
ZK_CONST_FE(fe_D_MINUS_ONE_SQ, ZK_FE_D_MINUS_ONE_SQ)
ZK_CONST_FE(fe_SQRT_AD_MINUS_ONE, ZK_FE_SQRT_AD_MINUS_ONE)
ZK_CONST_FE(fe_BASE_X, ZK_FE_BASE_X)
ZK_CONST_FE(fe_BASE_Y, ZK_FE_BASE_Y)
ZK_CONST_FE(fe_BASE_T, ZK_FE_BASE_T)

ZK_HD void ge_identity(ge& p) {
  p.X = fe_zero(); p.Y = fe_one(); p.Z = fe_one(); p.T = fe_zero();
}

ZK_HD void niels_identity(ge_niels& q) {
  q.ypx = fe_one(); q.ymx = fe_one(); q.xy2d = fe_zero();
}

// r = p + q (negate: r = p - q), q affine Niels.  7M.
ZK_HD void ge_madd(ge& r, const ge& p, const ge_niels& q, bool negate) {
  fe a, b, c, d, e, f, g, h, t0, t1, qa = q.ymx, qb = q.ypx;
  fe_cswap(qa, qb, negate);         // -q = (ymx, ypx, -xy2d)
  fe_sub(t0, p.Y, p.X);
  fe_add(t1, p.Y, p.X);
  fe_mul(a, t0, qa);
  fe_mul(b, t1, qb);
  fe_mul(c, p.T, q.xy2d);
  fe_add(d, p.Z, p.Z);
  fe_sub(e, b, a);
  fe_add(h, b, a);
  fe_sub_c(f, d, c);                // d is loose: carry
  fe_add(g, d, c);
  fe_cswap(f, g, negate);
  fe_mul(r.X, e, f);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, f, g);
  fe_mul(r.T, e, h);
}

// r = p + q, both extended.  9M.
ZK_HD void ge_add(ge& r, const ge& p, const ge& q) {
  fe a, b, c, d, e, f, g, h, t0, t1;
  fe_sub(t0, p.Y, p.X);
  fe_sub(t1, q.Y, q.X);
  fe_mul(a, t0, t1);
  fe_add(t0, p.Y, p.X);
  fe_add(t1, q.Y, q.X);
  fe_mul(b, t0, t1);
  fe_mul(c, p.T, q.T);
  fe_mul(c, c, fe_D2());
  fe_mul(d, p.Z, q.Z);
  fe_add(d, d, d);
  fe_sub(e, b, a);
  fe_add(h, b, a);
  fe_sub_c(f, d, c);
  fe_add(g, d, c);
  fe_mul(r.X, e, f);
  fe_mul(r.Y, g, h);
  fe_mul(r.Z, f, g);
  fe_mul(r.T, e, h);
}

// r = 2p.  4S + 4M (3M when the caller does not need T).
template <bool NEED_T = true>
ZK_HD void ge_double(ge& r, const ge& p) {
  fe A, B, C, E, F, G, H, t;
  fe_sq(A, p.X);
  fe_sq(B, p.Y);
  fe_sq(C, p.Z);
  fe_add(C, C, C);            // loose
  fe_add(t, p.X, p.Y);
  fe_sq(t, t);
  // Only what becomes the SECOND operand of a product (F, -(A+B)) is carried: fe_mul multiplies its second operand by 19 and
  // needs it below 2^27.7; its first operand may be as loose as 1.25 * 2^28 (a column then stays below 2^61.5).  E and G are
  // first operands only (round 5: three carry passes fewer per doubling; the same form as quad.hpp's quad_double).
  fe_add(H, A, B);            // loose: A + B < 2^27
  fe_sub4_loose(E, t, H);     // E = (X+Y)^2 - A - B + 4p, not carried
  fe_sub(G, B, A);            // G = B - A + 2p, not carried
  fe_sub_c(F, G, C);          // F = G - C, carried
  const fe zero = fe_zero();
  fe_sub_c(H, zero, H);       // H = -(A + B), carried
  fe_mul(r.X, E, F);
  fe_mul(r.Y, G, H);
  fe_mul(r.Z, G, F);
  if (NEED_T) fe_mul(r.T, E, H);
}

ZK_HD void ge_neg(ge& r, const ge& p) {
  fe_neg(r.X, p.X);
  r.Y = p.Y;
  r.Z = p.Z;
  fe_neg(r.T, p.T);
}

// identity of the ristretto group <=> the representative lies in E[4]
// <=> X == 0 or Y == 0  (RFC 9496 sec 4.3.3 against (0,1,1,0))
ZK_HD bool ge_is_identity(const ge& p) { return fe_is_zero(p.X) | fe_is_zero(p.Y); }

// SQRT_RATIO_M1(u, v), RFC 9496 sec 4.2
ZK_HD bool fe_sqrt_ratio_m1(fe& r, const fe& u, const fe& v) {
  fe v3, v7, t, check, neg_u, neg_u_i, r_prime;
  fe_sq(t, v);
  fe_mul(v3, t, v);
  fe_sq(t, v3);
  fe_mul(v7, t, v);
  fe_mul(t, u, v7);
  fe_pow22523(t, t);
  fe_mul(t, t, v3);
  fe_mul(r, t, u);
  fe_sq(t, r);
  fe_mul(check, t, v);
  fe_neg(neg_u, u);
  fe_mul(neg_u_i, neg_u, fe_SQRT_M1());
  const bool correct_sign = fe_eq(check, u);
  const bool flipped_sign = fe_eq(check, neg_u);
  const bool flipped_sign_i = fe_eq(check, neg_u_i);
  fe_mul(r_prime, r, fe_SQRT_M1());
  fe_cmov(r, r_prime, flipped_sign | flipped_sign_i);
  fe neg_r;
  fe_neg(neg_r, r);
  fe_cmov(r, neg_r, fe_is_negative(r));
  return correct_sign | flipped_sign;
}

// DECODE (RFC 9496 sec 4.3.1): 32 bytes as 8 little-endian words -> affine (x, y).
// Returns false for every encoding the RFC rejects (non-canonical, negative,
// non-square, negative t, y = 0).
ZK_HD bool ristretto_decode_affine(fe& x, fe& y, const uint32_t w[8]) {
  fe s, ss, u1, u2, u2_sqr, v, t, invsqrt, den_x, den_y;
  fe_from_words(s, w);
  uint32_t chk[8];
  fe_to_words(chk, s);
  bool canonical = true;
  ZK_UNROLL for (int i = 0; i < 8; ++i) canonical &= (chk[i] == w[i]);
  const bool negative = w[0] & 1;
  fe_sq(ss, s);
  fe_sub(u1, fe_one(), ss);
  fe_add(u2, fe_one(), ss);
  fe_sq(u2_sqr, u2);
  fe_sq(t, u1);
  fe_mul(t, t, fe_D());
  fe_add(t, t, u2_sqr);                 // loose
  fe_sub_c(v, fe_zero(), t);            // v = -(D u1^2) - u2_sqr
  fe_mul(t, v, u2_sqr);
  const bool was_square = fe_sqrt_ratio_m1(invsqrt, fe_one(), t);
  fe_mul(den_x, invsqrt, u2);
  fe_mul(den_y, invsqrt, den_x);
  fe_mul(den_y, den_y, v);
  fe_mul(x, s, den_x);
  fe_add(x, x, x);
  fe_carry(x);
  fe nx;
  fe_neg(nx, x);
  fe_cmov(x, nx, fe_is_negative(x));
  fe_mul(y, u1, den_y);
  fe_mul(t, x, y);
  return canonical & !negative & was_square & !fe_is_negative(t) & !fe_is_zero(y);
}

ZK_HD bool ristretto_decode(ge& p, const uint32_t w[8]) {
  const bool ok = ristretto_decode_affine(p.X, p.Y, w);
  p.Z = fe_one();
  fe_mul(p.T, p.X, p.Y);
  return ok;
}

ZK_HD void niels_from_affine(ge_niels& q, const fe& x, const fe& y) {
  fe t;
  fe_add_c(q.ypx, y, x);
  fe_sub_c(q.ymx, y, x);
  fe_mul(t, x, y);
  fe_mul(q.xy2d, t, fe_D2());
}

// ENCODE (RFC 9496 sec 4.3.2)
ZK_HD void ristretto_encode(uint32_t out[8], const ge& p) {
  fe u1, u2, t, invsqrt, den1, den2, z_inv, ix0, iy0, enchanted, x, y, den_inv, s, ny, ns;
  fe_add(u1, p.Z, p.Y);
  fe_sub(t, p.Z, p.Y);
  fe_mul(u1, u1, t);
  fe_mul(u2, p.X, p.Y);
  fe_sq(t, u2);
  fe_mul(t, t, u1);
  (void)fe_sqrt_ratio_m1(invsqrt, fe_one(), t);
  fe_mul(den1, invsqrt, u1);
  fe_mul(den2, invsqrt, u2);
  fe_mul(z_inv, den1, den2);
  fe_mul(z_inv, z_inv, p.T);
  fe_mul(ix0, p.X, fe_SQRT_M1());
  fe_mul(iy0, p.Y, fe_SQRT_M1());
  fe_mul(enchanted, den1, fe_INVSQRT_A_MINUS_D());
  fe_mul(t, p.T, z_inv);
  const bool rotate = fe_is_negative(t);
  x = p.X; y = p.Y; den_inv = den2;
  fe_cmov(x, iy0, rotate);
  fe_cmov(y, ix0, rotate);
  fe_cmov(den_inv, enchanted, rotate);
  fe_mul(t, x, z_inv);
  fe_neg(ny, y);
  fe_cmov(y, ny, fe_is_negative(t));
  fe_sub(t, p.Z, y);
  fe_mul(s, den_inv, t);
  fe_neg(ns, s);
  fe_cmov(s, ns, fe_is_negative(s));
  fe_to_words(out, s);
}

// MAP (RFC 9496 sec 4.3.4) and element derivation from 64 uniform bytes
ZK_HD void ristretto_elligator(ge& p, const fe& t0) {
  fe r, u, v, s, s_prime, c, n, w0, w1, w2, w3, t, ss, ns;
  fe_sq(r, t0);
  fe_mul(r, r, fe_SQRT_M1());
  fe_add(u, r, fe_one());
  fe_mul(u, u, fe_ONE_MINUS_D_SQ());
  fe_mul(t, r, fe_D());
  fe_add(t, t, fe_one());
  fe_sub_c(t, fe_zero(), t);        // -1 - r d
  fe_add(v, r, fe_D());
  fe_mul(v, t, v);
  const bool was_square = fe_sqrt_ratio_m1(s, u, v);
  fe_mul(s_prime, s, t0);
  fe_neg(ns, s_prime);
  fe_cmov(s_prime, ns, !fe_is_negative(s_prime));   // -|s t|
  fe_neg(c, fe_one());
  fe_cmov(s, s_prime, !was_square);
  fe_cmov(c, r, !was_square);
  fe_sub(t, r, fe_one());
  fe_mul(n, c, t);
  fe_mul(n, n, fe_D_MINUS_ONE_SQ());
  fe_sub(n, n, v);
  fe_mul(w0, s, v);
  fe_add(w0, w0, w0);
  fe_mul(w1, n, fe_SQRT_AD_MINUS_ONE());
  fe_sq(ss, s);
  fe_sub(w2, fe_one(), ss);
  fe_add(w3, fe_one(), ss);
  fe_mul(p.X, w0, w3);
  fe_mul(p.Y, w2, w1);
  fe_mul(p.Z, w1, w3);
  fe_mul(p.T, w0, w2);
}

ZK_HD void ristretto_from_uniform_words(ge& p, const uint32_t w[16]) {
  fe t1, t2;
  ge p1, p2;
  fe_from_words(t1, w);
  fe_from_words(t2, w + 8);
  ristretto_elligator(p1, t1);
  ristretto_elligator(p2, t2);
  ge_add(p, p1, p2);
}

}  // namespace zk
