// quad.hpp -- point doubling / addition spread over a quad of lanes (DPP).
//
// A variable-base MSM ends in a chain of ~255 dependent doublings per result
// (Horner over the windows).  On one lane that chain takes ~0.7 ms on gfx950
// whatever the batch size, and a 1024-transaction batch offers only 1024 such
// chains -- 16 wavefronts on a chip with 1024 SIMDs.  The chain cannot be
// shortened, but each link can: a doubling is four independent squarings
// followed by four independent multiplications.  Here four adjacent lanes (a
// DPP "quad") each keep a full copy of the point, each computes ONE of the four
// field operations of a stage, and the results are exchanged with quad_perm DPP
// moves (register-to-register, no LDS).  Depth per doubling: 1 squaring + 1
// multiplication instead of 4 + 4.
#pragma once
#include "curve.hpp"

namespace zk {

// broadcast lane K of every quad to the four lanes of that quad
template <int K>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
  constexpr int ctrl = K * 0x55;   // quad_perm:[K,K,K,K]
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, ctrl, 0xf, 0xf, false);
}

template <int K>
__device__ __forceinline__ void quad_bcast_fe(fe& out, const fe& v) {
#pragma unroll
  for (int i = 0; i < 10; ++i) out.v[i] = quad_bcast<K>(v.v[i]);
}

__device__ __forceinline__ void fe_select4(fe& out, int r, const fe& a0, const fe& a1, const fe& a2, const fe& a3) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    uint32_t lo = (r & 1) ? a1.v[i] : a0.v[i];
    uint32_t hi = (r & 1) ? a3.v[i] : a2.v[i];
    out.v[i] = (r & 2) ? hi : lo;
  }
}

// out = pick ? b : a
__device__ __forceinline__ void fe_select2(fe& out, bool pick, const fe& a, const fe& b) {
#pragma unroll
  for (int i = 0; i < 10; ++i) out.v[i] = pick ? b.v[i] : a.v[i];
}

// p <- 2p, all four lanes of the quad hold (and receive) the same point.  r = lane & 3.
// Between the two stages only what becomes a SECOND operand of a product (F, -(A+B)) is carried; E and G are first operands
// only (X3 = E F, Y3 = G H', Z3 = G F, T3 = E H'), and stay as the sums they are.  Bounds: first operand < 1.25 * 2^28 (even
// limbs) / 2^27 (odd), second tight: a column of fe_mul is at most 5 * 19 * 2^54.4 + 5 * 19 * 2^53.4 < 2^61.5 (r05: three
// carry passes and forty conditional moves fewer per doubling than the form that carried E, G and A+B and chose among four).
__device__ __forceinline__ void quad_double(ge& p, int r) {
  fe in, xy, v;
  fe_add(xy, p.X, p.Y);                       // loose
  fe_select4(in, r, p.X, p.Y, p.Z, xy);
  fe_sq(v, in);                               // lane r: X^2, Y^2, Z^2, (X+Y)^2
  fe A, B, C, t, E, F, G, Hs, Hn;
  quad_bcast_fe<0>(A, v);
  quad_bcast_fe<1>(B, v);
  quad_bcast_fe<2>(C, v);
  quad_bcast_fe<3>(t, v);
  fe_add(C, C, C);                            // 2 Z^2 (loose)
  fe_add(Hs, A, B);                           // loose
  fe_sub4_loose(E, t, Hs);                    // (X+Y)^2 - A - B, not carried
  fe_sub(G, B, A);                            // B - A + 2p, not carried
  fe_sub_c(F, G, C);                          // G - 2 Z^2, carried
  const fe zero = fe_zero();
  fe_sub_c(Hn, zero, Hs);                     // -(A+B), carried
  fe a, b, m;
  fe_select2(a, r == 1 || r == 2, E, G);      // X3 = E F, Y3 = G H', Z3 = G F, T3 = E H'
  fe_select2(b, (r & 1) != 0, F, Hn);
  fe_mul(m, a, b);
  quad_bcast_fe<0>(p.X, m);
  quad_bcast_fe<1>(p.Y, m);
  quad_bcast_fe<2>(p.Z, m);
  quad_bcast_fe<3>(p.T, m);
}

// p <- p + q (both extended, identical on the four lanes).  Depth: 3 multiplications.
__device__ __forceinline__ void quad_add(ge& p, const ge& q, int r) {
  fe s1, s2, d1, d2, a, b, m;
  fe_sub(d1, p.Y, p.X);
  fe_sub(d2, q.Y, q.X);
  fe_add(s1, p.Y, p.X);
  fe_add(s2, q.Y, q.X);
  fe_select4(a, r, d1, s1, p.T, p.Z);
  fe_select4(b, r, d2, s2, q.T, q.Z);
  fe_mul(m, a, b);                            // lane r: A, B, T1 T2, Z1 Z2
  fe A, B, C, D, E, F, G, H;
  quad_bcast_fe<0>(A, m);
  quad_bcast_fe<1>(B, m);
  quad_bcast_fe<2>(C, m);
  quad_bcast_fe<3>(D, m);
  fe_mul(C, C, fe_D2());                      // same on all lanes (redundant, keeps the code uniform)
  fe_add(D, D, D);
  fe_sub(E, B, A);
  fe_add(H, B, A);
  fe_sub_c(F, D, C);
  fe_add(G, D, C);
  fe_select2(a, r == 1 || r == 2, E, G);      // X3 = E F, Y3 = G H, Z3 = G F, T3 = E H: G (loose) is a first operand only
  fe_select2(b, (r & 1) != 0, F, H);
  fe_mul(m, a, b);
  quad_bcast_fe<0>(p.X, m);
  quad_bcast_fe<1>(p.Y, m);
  quad_bcast_fe<2>(p.Z, m);
  quad_bcast_fe<3>(p.T, m);
}

}  // namespace zk
