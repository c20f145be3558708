// keccak_coop.hpp -- Keccak-f[1600] with ONE state spread over the lanes of a wavefront.
//
// Why: the Merlin transcript of a proof is ~36 dependent Keccak-f permutations.  With one state per
// lane (merlin_dev.hpp) a permutation is ~4300 dependent VALU instructions at 4 cycles each, whatever
// the batch size -- 0.5 ms for the transcript of a 1024-transaction batch that keeps 16 wavefronts of
// a 1024-SIMD chip busy.  Here the 25 64-bit words of a state live in 25 lanes, a round is ~36
// instructions per lane, and a 1024-transaction batch is 1024 wavefronts (prep_kernels.hpp,
// k_transcript_coop).  (FIPS 202 sec 3.2.)
//
// Lane layout (lane = 8 y + pos; DPP rows are 16 lanes = two y):
//     pos     0   1   2   3   4   5   6   7
//     x       4   0   1   2   3   4   0   -        pos 1..5 hold A[x][y]; pos 0 and 6 are copies of
//                                                  x = 4 and x = 0 so that "x - 1" and "x + 1" are plain
//                                                  DPP row shifts; pos 7 and y >= 5 are dead (kept zero)
// Each lane keeps its word as (lo, hi) 32-bit halves.  One round:
//   theta  column parity = XOR over y: one DPP row_ror:8 inside a row, then three row swaps
//          (v_permlane16_swap / v_permlane32_swap, gfx950) combine the rows for both halves at once;
//          C[x-1], C[x+1] by row_shr:1 / row_shl:1; rot(.,1) two v_alignbit; apply: one v_bitop3 per half
//   rho    lane-dependent 64-bit rotation: conditional swap of the halves + two v_alignbit
//   pi+chi every lane (copies included) GATHERS the three words of its row it needs,
//          B[x][y], B[x+1][y], B[x+2][y] with B = pi(rho(theta(A))), by ds_bpermute (LDS crossbar, no
//          memory), then one v_bitop3 per half; the copies are thereby refreshed for free
//   iota   v_bitop3 with a lane mask.
//
// The algorithm is written once over a traits class so that the very same code runs on the host with
// emulated cross-lane primitives (tests/test_host_logic.py via hostlib.cpp) and on the device with the
// real ones; tests on the GPU check the device result against the oracle's Keccak.
#pragma once
#include <stdint.h>

#include "field.hpp"   // ZK_HD

#if defined(__HIPCC__) || defined(__HIP__)
#define ZK_HD_INL __host__ __device__ __forceinline__
#else
#define ZK_HD_INL inline
#endif

namespace zk {
namespace coop {

struct KcLane {          // constants of one lane
  uint32_t live;         // ~0 on the 35 lanes that carry state (5 y x 7 pos), 0 on dead lanes
  uint32_t q;            // state word x + 5 y held by this lane (dead lanes: 0, never used)
  uint32_t primary;      // ~0 on THE holder of word q (pos 1..5)
  uint32_t rot_swap;     // rho as a right rotation by t = (64 - r) & 63: ~0 when t >= 32 (swap the halves first)
  uint32_t rot_t;        // t & 31
  uint32_t src[3];       // ds_bpermute addresses (4 x lane) of the sources of B[x][y], B[x+1][y], B[x+2][y]
  uint32_t iota;         // ~0 on the holders of word 0
};

ZK_HD KcLane kc_lane(uint32_t lane) {
  const int KC_RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  KcLane c;
  const uint32_t y = lane >> 3, pos = lane & 7;
  const bool live = y < 5 && pos < 7;
  const uint32_t x = (pos + 4) % 5;
  c.live = live ? ~0u : 0u;
  c.q = live ? x + 5 * y : 0;
  c.primary = (live && pos >= 1 && pos <= 5) ? ~0u : 0u;
  const uint32_t r = live ? (uint32_t)KC_RHO[x + 5 * y] : 0, t = (64 - r) & 63;
  c.rot_swap = t >= 32 ? ~0u : 0u;
  c.rot_t = t & 31;
  for (uint32_t d = 0; d < 3; ++d) {
    // destination (X, Y) = (x + d, y) receives rho(theta(A))[sx][sy] with (X, Y) = (sy, 2 sx + 3 sy):
    // sy = X, sx = (Y - 3 X) / 2 = 3 Y + X (mod 5)
    const uint32_t X = (x + d) % 5, Y = y;
    const uint32_t sy = X, sx = (3 * Y + X) % 5;
    c.src[d] = live ? 4 * (8 * sy + sx + 1) : 4 * lane;
  }
  c.iota = (live && x == 0 && y == 0) ? ~0u : 0u;
  return c;
}

// T supplies: type V (a 32-bit value per lane); V-valued constants of the lane; and
//   xor3(a,b,c), chi(a,b,c) = a ^ (~b & c), xor_and(a,b,m) = a ^ (b & m), and_(a,m), sel(m,a,b) = m ? a : b,
//   alignbit(hi, lo, s) = low 32 bits of ((hi:lo) >> (s & 31)) with a per-lane s,
//   ror8(v), shr1(v) (lane i reads i-1 inside its row of 16; row-lane 0 keeps its own value), shl1(v),
//   swap16(a, b) (rows 1,3 of a <-> rows 0,2 of b), swap32(a, b) (rows 2,3 of a <-> rows 0,1 of b),
//   gather(addr, v) (lane i reads v of lane addr_i / 4), splat(u32).
template <class T>
struct KeccakCoop {
  using V = typename T::V;
  struct Consts { V live, rot_swap, rot_t, src0, src1, src2, iota; };

  static ZK_HD_INL void round(V& lo, V& hi, const Consts& c, uint32_t rc_lo, uint32_t rc_hi) {
    // theta: column parities over y
    V pl = T::xor2(lo, T::ror8(lo)), ph = T::xor2(hi, T::ror8(hi));
    V t = pl, u = ph;
    T::swap16(t, u);                   // t = [lo0 hi0 lo2 hi2], u = [lo1 hi1 lo3 hi3] (rows)
    V s = T::xor2(t, u), s2 = s;
    T::swap32(s, s2);                  // s = [lo01 hi01 lo01 hi01], s2 = [lo23 hi23 lo23 hi23]
    V S = T::xor2(s, s2);              // [L H L H]
    V cl = S, ch = S;
    T::swap16(cl, ch);                 // cl = L, ch = H in every row
    const V ml = T::shr1(cl), mh = T::shr1(ch);         // C[x-1]
    const V nl = T::shl1(cl), nh = T::shl1(ch);         // C[x+1]
    const V thirty_one = T::splat(31);
    const V rl = T::alignbit(nl, nh, thirty_one);       // rot(C[x+1], 1): lo' = (lo << 1) | (hi >> 31)
    const V rh = T::alignbit(nh, nl, thirty_one);
    lo = T::xor3(lo, ml, rl);
    hi = T::xor3(hi, mh, rh);
    // rho: rotate right by t (= left by r)
    const V a0 = T::sel(c.rot_swap, hi, lo), a1 = T::sel(c.rot_swap, lo, hi);
    const V bl = T::alignbit(a1, a0, c.rot_t), bh = T::alignbit(a0, a1, c.rot_t);
    // pi + chi: gather the three words of the row
    const V g0l = T::gather(c.src0, bl), g0h = T::gather(c.src0, bh);
    const V g1l = T::gather(c.src1, bl), g1h = T::gather(c.src1, bh);
    const V g2l = T::gather(c.src2, bl), g2h = T::gather(c.src2, bh);
    lo = T::chi(g0l, g1l, g2l);
    hi = T::chi(g0h, g1h, g2h);
    // iota, and keep the dead lanes at zero (they feed the column parities)
    lo = T::and_(T::xor_and(lo, T::splat(rc_lo), c.iota), c.live);
    hi = T::and_(T::xor_and(hi, T::splat(rc_hi), c.iota), c.live);
  }

  static ZK_HD_INL void permute(V& lo, V& hi, const Consts& c) {
    const uint32_t KC_RC_LO[24] = {0x00000001u, 0x00008082u, 0x0000808Au, 0x80008000u, 0x0000808Bu, 0x80000001u, 0x80008081u, 0x00008009u,
                                   0x0000008Au, 0x00000088u, 0x80008009u, 0x8000000Au, 0x8000808Bu, 0x0000008Bu, 0x00008089u, 0x00008003u,
                                   0x00008002u, 0x00000080u, 0x0000800Au, 0x8000000Au, 0x80008081u, 0x00008080u, 0x80000001u, 0x80008008u};
    const uint32_t KC_RC_HI[24] = {0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u,
                                   0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u,
                                   0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u, 0x80000000u, 0x80000000u, 0x00000000u, 0x80000000u};
    ZK_NOUNROLL for (int r = 0; r < 24; ++r) round(lo, hi, c, KC_RC_LO[r], KC_RC_HI[r]);
  }
};

// ---- host emulation of a 64-lane wavefront --------------------------------------------------
struct LaneVec { uint32_t l[64]; };

struct HostTraits {
  using V = LaneVec;
  template <class F> static V map1(const V& a, F f) { V r; for (int i = 0; i < 64; ++i) r.l[i] = f(a.l[i]); return r; }
  template <class F> static V map2(const V& a, const V& b, F f) { V r; for (int i = 0; i < 64; ++i) r.l[i] = f(a.l[i], b.l[i]); return r; }
  template <class F> static V map3(const V& a, const V& b, const V& c, F f) { V r; for (int i = 0; i < 64; ++i) r.l[i] = f(a.l[i], b.l[i], c.l[i]); return r; }
  static V splat(uint32_t x) { V r; for (int i = 0; i < 64; ++i) r.l[i] = x; return r; }
  static V xor2(const V& a, const V& b) { return map2(a, b, [](uint32_t x, uint32_t y) { return x ^ y; }); }
  static V xor3(const V& a, const V& b, const V& c) { return map3(a, b, c, [](uint32_t x, uint32_t y, uint32_t z) { return x ^ y ^ z; }); }
  static V chi(const V& a, const V& b, const V& c) { return map3(a, b, c, [](uint32_t x, uint32_t y, uint32_t z) { return x ^ (~y & z); }); }
  static V xor_and(const V& a, const V& b, const V& m) { return map3(a, b, m, [](uint32_t x, uint32_t y, uint32_t z) { return x ^ (y & z); }); }
  static V and_(const V& a, const V& m) { return map2(a, m, [](uint32_t x, uint32_t y) { return x & y; }); }
  static V sel(const V& m, const V& a, const V& b) { return map3(m, a, b, [](uint32_t mm, uint32_t x, uint32_t y) { return mm ? x : y; }); }
  static V alignbit(const V& hi, const V& lo, const V& s) {
    return map3(hi, lo, s, [](uint32_t h, uint32_t l, uint32_t sh) { return (uint32_t)((((uint64_t)h << 32) | l) >> (sh & 31)); });
  }
  static V ror8(const V& v) { V r; for (int i = 0; i < 64; ++i) r.l[i] = v.l[(i & ~15) | ((i + 8) & 15)]; return r; }
  static V shr1(const V& v) { V r; for (int i = 0; i < 64; ++i) r.l[i] = (i & 15) ? v.l[i - 1] : v.l[i]; return r; }
  static V shl1(const V& v) { V r; for (int i = 0; i < 64; ++i) r.l[i] = ((i & 15) != 15) ? v.l[i + 1] : v.l[i]; return r; }
  static void swap16(V& a, V& b) {
    for (int row = 1; row < 4; row += 2)
      for (int i = 0; i < 16; ++i) { const uint32_t t = a.l[16 * row + i]; a.l[16 * row + i] = b.l[16 * (row - 1) + i]; b.l[16 * (row - 1) + i] = t; }
  }
  static void swap32(V& a, V& b) {
    for (int i = 0; i < 32; ++i) { const uint32_t t = a.l[32 + i]; a.l[32 + i] = b.l[i]; b.l[i] = t; }
  }
  static V gather(const V& addr, const V& v) { V r; for (int i = 0; i < 64; ++i) r.l[i] = v.l[(addr.l[i] >> 2) & 63]; return r; }
};

inline KeccakCoop<HostTraits>::Consts host_consts() {
  KeccakCoop<HostTraits>::Consts c;
  for (uint32_t i = 0; i < 64; ++i) {
    const KcLane k = kc_lane(i);
    c.live.l[i] = k.live; c.rot_swap.l[i] = k.rot_swap; c.rot_t.l[i] = k.rot_t;
    c.src0.l[i] = k.src[0]; c.src1.l[i] = k.src[1]; c.src2.l[i] = k.src[2]; c.iota.l[i] = k.iota;
  }
  return c;
}

// Keccak-f[1600] of a lane-major state through the emulated wavefront (for the CPU tests)
inline void keccak_f1600_emulated(uint64_t s[25]) {
  LaneVec lo, hi;
  for (uint32_t i = 0; i < 64; ++i) {
    const KcLane k = kc_lane(i);
    lo.l[i] = k.live ? (uint32_t)s[k.q] : 0;
    hi.l[i] = k.live ? (uint32_t)(s[k.q] >> 32) : 0;
  }
  const auto c = host_consts();
  KeccakCoop<HostTraits>::permute(lo, hi, c);
  for (uint32_t i = 0; i < 64; ++i) {
    const KcLane k = kc_lane(i);
    if (k.primary) s[k.q] = (uint64_t)lo.l[i] | ((uint64_t)hi.l[i] << 32);
  }
}

}  // namespace coop
}  // namespace zk
