// prover_dev.hpp -- bulletproofs::r1cs::Prover for a constraint system described as data, with EVERYTHING
// between the multiscalar multiplications on the device (BASELINE.json configs[4];
// upstream `r1cs::Prover::{commit, prove}`, `InnerProductProof::create`, merlin's `TranscriptRng` -- sources not
// mounted; byte-level behaviour follows r1cs_prover.hpp / oracle/r1cs.c, and the tests require byte-identical
// proofs from all three).
//
// One workgroup per proof.  A proof is a sequence of PHASES separated by the multiscalar multiplications over the
// generator tables (value commitments | A_I1 A_O1 S1 | A_I2 A_O2 S2 | T_1 T_3..T_6 | L_j R_j per round); a phase
// absorbs the points of the previous multiplication into the proof's Merlin transcript (thread 0: STROBE is a
// strictly serial byte machine), draws what it needs from the transcript and from the TranscriptRng, does the
// vector algebra with all threads and writes the scalars of the next multiplication as canonical words.  The
// host only queues kernels: no byte of a proof in the making crosses PCIe.
//
// The phase functions are written against an Env (thread id, barrier, sum over the workgroup, a 52-word scratch
// for the STROBE state) so that the very same code runs on the host with one "thread" -- hostlib.cpp proves with
// it on the CPU and the tests compare the bytes with the host prover and the oracle.
//
// Scalars in the per-proof state are scm (Montgomery form, canonical range, 8 words); arithmetic between loads and
// stores is in the lazy limb form scl (sc_dev.hpp).
#pragma once
#include "sc_dev.hpp"
#if defined(__HIPCC__) || defined(__HIP__)
#include "merlin_dev.hpp"
#endif
#if !defined(__HIP_DEVICE_COMPILE__)
#include "keccak.hpp"
#endif
#include "keccak_coop.hpp"

namespace zk {

constexpr uint32_t PV_GIVEN = 0xffffffffu;
constexpr uint32_t PV_MAX_LABEL = 31;

struct PvShape {
  uint32_t m, n1, n, pn, k, n_cons, n_chal2, n_mono, gens_capacity;
  uint32_t two_phase;                 // "r1cs-2phase" (randomized constraints) or "r1cs-1phase"
  uint32_t n_given;                   // (left, right) pairs handed in per proof
  uint32_t proof_len, proof_stride;   // bytes
  // per-proof state, offsets in words
  uint32_t o_tr, o_rng, o_v, o_vbl, o_aL, o_aR, o_aO, o_sL, o_sR, o_blind, o_chal, o_c2, o_sym, o_wL, o_wR, o_wO, o_wV,
      o_t, o_tb, o_zpow, o_ypow, o_yinv, o_flag, state_words;
  // rows of the multiscalar multiplications, per proof: first row and terms
  uint32_t r1_terms, r2_terms;        // scalars per proof in the phase-1 / phase-2 commitment rows
};
// challenge slots (o_chal + 8 * slot)
enum { PV_Y = 0, PV_Z, PV_U, PV_X, PV_W, PV_YINV, PV_CHAL_SLOTS };
// blinding slots (o_blind + 8 * slot)
enum { PV_IBL1 = 0, PV_OBL1, PV_SBL1, PV_IBL2, PV_OBL2, PV_SBL2, PV_BLIND_SLOTS };

struct PvPlan {   // constant per statement shape; device (or host) pointers
  const uint32_t* init;          // 52 words: STROBE state after Transcript::new(label) + "dom-sep" "r1cs v1"
  const uint8_t* chal_labels;    // n_chal2 x 32 bytes: length, then the label
  const uint32_t* mono_chal;     // n_mono (monomial 0 = the constant 1)
  const uint32_t* mono_pow;
  const uint32_t* con_off;       // n_cons + 1: the constraints in the order the system emitted them
  const uint32_t* t_kind;        // per term: 0 committed, 1 / 2 / 3 left / right / out of a multiplier, 4 the constant one
  const uint32_t* t_idx;
  const uint32_t* t_mono;
  const uint32_t* t_coef;        // 8 words per term, Montgomery form
  const uint32_t* mult_def;      // 2 n: defining constraints of (left, right), or PV_GIVEN
  const uint32_t* given_slot;    // n: position in the proof's given list, or PV_GIVEN
  // the flattening plan of cloak_plan.hpp (targets wL | wR | wO | wV | wc)
  const uint32_t* tgt_off;
  const uint32_t* term_info;
  const uint32_t* prod_qm;       // (q, monomial) pairs
  const uint32_t* prod_coef;     // 10 limbs
};

struct PvBatch {   // per batch; device (or host) pointers
  uint32_t* state;               // batch x state_words
  const uint32_t* values;        // batch x m x 8 canonical words (plain)
  const uint32_t* blindings;     // batch x m x 8
  const uint32_t* given;         // batch x n_given x 16
  const uint32_t* rng_seed;      // batch x 8: SHAKE256(seed || "rng" || LE64(0))
  uint8_t* proofs;               // batch x proof_stride
  uint32_t* rows0;               // value commitments: batch x m rows of 2 scalars
  uint32_t* rows1;               // batch x r1_terms scalars (A_I1 | A_O1 | S1)
  uint32_t* rows2;               // batch x r2_terms
  uint32_t* rows3;               // batch x 5 rows of 2 scalars
  uint32_t* ipa_lv; uint32_t* ipa_rv; uint32_t* ipa_cg; uint32_t* ipa_ch; uint32_t* ipa_w; uint32_t* ipa_u;
};

// ---- STROBE-128 / Merlin on a 52-word buffer (50 state words, position, begin marker) ----------------------
struct PvStrobe {
  uint32_t* w;
  static constexpr unsigned kRate = 166;
  ZK_HD unsigned pos() const { return w[50]; }
  ZK_HD void xor_in(unsigned i, uint32_t b) { w[i >> 2] ^= b << (8 * (i & 3)); }
  ZK_HD uint32_t get(unsigned i) const { return (w[i >> 2] >> (8 * (i & 3))) & 0xffu; }
  ZK_HD void clear(unsigned i) { w[i >> 2] &= ~(0xffu << (8 * (i & 3))); }
  ZK_HD_NOINLINE void permute() {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t lo[25], hi[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) { lo[i] = w[2 * i]; hi[i] = w[2 * i + 1]; }
    keccak_f1600_halves(lo, hi);
#pragma unroll
    for (int i = 0; i < 25; ++i) { w[2 * i] = lo[i]; w[2 * i + 1] = hi[i]; }
#else
    uint64_t s[25];
    for (int i = 0; i < 25; ++i) s[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    keccak_f1600(s);
    for (int i = 0; i < 25; ++i) { w[2 * i] = (uint32_t)s[i]; w[2 * i + 1] = (uint32_t)(s[i] >> 32); }
#endif
  }
  ZK_HD_NOINLINE void run_f() {
    xor_in(w[50], w[51] & 0xffu);
    xor_in(w[50] + 1, 0x04);
    xor_in(kRate + 1, 0x80);
    permute();
    w[50] = 0;
    w[51] = 0;
  }
  ZK_HD void absorb_byte(uint32_t b) {
    xor_in(w[50], b);
    if (++w[50] == kRate) run_f();
  }
  ZK_HD void begin_op(uint32_t flags) {
    const uint32_t old_begin = w[51] & 0xffu;
    w[51] = w[50] + 1;
    absorb_byte(old_begin);
    absorb_byte(flags);
    if ((flags & (4u | 32u)) && w[50] != 0) run_f();
  }
  // flags: I 1, A 2, C 4, T 8, M 16, K 32
  ZK_HD_NOINLINE void meta_ad(const uint8_t* d, unsigned n, bool more) { if (!more) begin_op(16u | 2u); for (unsigned i = 0; i < n; ++i) absorb_byte(d[i]); }
  ZK_HD_NOINLINE void ad(const uint8_t* d, unsigned n) { begin_op(2u); for (unsigned i = 0; i < n; ++i) absorb_byte(d[i]); }
  ZK_HD_NOINLINE void key(const uint8_t* d, unsigned n) {
    begin_op(2u | 4u);
    for (unsigned i = 0; i < n; ++i) { clear(w[50]); xor_in(w[50], d[i]); if (++w[50] == kRate) run_f(); }
  }
  ZK_HD_NOINLINE void prf(uint8_t* out, unsigned n) {
    begin_op(1u | 2u | 4u);
    for (unsigned i = 0; i < n; ++i) { out[i] = (uint8_t)get(w[50]); clear(w[50]); if (++w[50] == kRate) run_f(); }
  }
  // Merlin
  ZK_HD void le32(uint8_t b[4], uint32_t n) { b[0] = (uint8_t)n; b[1] = (uint8_t)(n >> 8); b[2] = (uint8_t)(n >> 16); b[3] = (uint8_t)(n >> 24); }
  ZK_HD_NOINLINE void append_message(const char* label, unsigned label_len, const uint8_t* msg, unsigned n) {
    uint8_t len[4];
    le32(len, n);
    meta_ad((const uint8_t*)label, label_len, false);
    meta_ad(len, 4, true);
    ad(msg, n);
  }
  ZK_HD void append_words(const char* label, unsigned label_len, const uint32_t* words, unsigned n_words) {
    uint8_t b[32];
    for (unsigned i = 0; i < n_words && i < 8; ++i) le32(b + 4 * i, words[i]);
    append_message(label, label_len, b, 4 * n_words);
  }
  ZK_HD void append_u64(const char* label, unsigned label_len, uint64_t x) {
    const uint32_t wv[2] = {(uint32_t)x, (uint32_t)(x >> 32)};
    append_words(label, label_len, wv, 2);
  }
  ZK_HD void challenge_wide(const char* label, unsigned label_len, uint32_t out[16]) {
    uint8_t len[4], b[64];
    le32(len, 64);
    meta_ad((const uint8_t*)label, label_len, false);
    meta_ad(len, 4, true);
    prf(b, 64);
    for (int i = 0; i < 16; ++i) out[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
  }
  ZK_HD_NOINLINE scm challenge_scalar(const char* label, unsigned label_len) {
    uint32_t wd[16];
    challenge_wide(label, label_len, wd);
    return scm_from_wide(wd);
  }
  // TranscriptRng
  ZK_HD void rekey_with_witness(const char* label, unsigned label_len, const uint32_t* words8) {
    uint8_t len[4], b[32];
    le32(len, 32);
    for (int i = 0; i < 8; ++i) le32(b + 4 * i, words8[i]);
    meta_ad((const uint8_t*)label, label_len, false);
    meta_ad(len, 4, true);
    key(b, 32);
  }
  ZK_HD void finalize_rng(const uint32_t* seed8) {
    uint8_t b[32];
    for (int i = 0; i < 8; ++i) le32(b + 4 * i, seed8[i]);
    meta_ad((const uint8_t*)"rng", 3, false);
    key(b, 32);
  }
  ZK_HD_NOINLINE scm rng_scalar() {
    uint8_t len[4], b[64];
    uint32_t wd[16];
    le32(len, 64);
    meta_ad(len, 4, false);
    prf(b, 64);
    for (int i = 0; i < 16; ++i) wd[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    return scm_from_wide(wd);
  }
};
// The TranscriptRng's draws, word-wise: once the generator is keyed every rng_fill(64) is the same STROBE sequence --
// meta-AD(LE32(64)), PRF(64) -- which XORs ten fixed bytes into the state at the current position (32 right after the
// keying, 64 from then on), pads, runs Keccak-f and hands out (and zeroes) the first 64 bytes.  With the state in
// registers a draw is one permutation and a reduction mod l: what the serial byte machine above needs ~200 LDS
// round trips for.  One lane per proof (k_pv_rng).
struct PvRng {
  uint32_t lo[25], hi[25];
  uint32_t pos, pos_begin;
  ZK_HD void load(const uint32_t* w) {
    ZK_UNROLL for (int i = 0; i < 25; ++i) { lo[i] = w[2 * i]; hi[i] = w[2 * i + 1]; }
    pos = w[50]; pos_begin = w[51];
  }
  ZK_HD void store(uint32_t* w) const {
    ZK_UNROLL for (int i = 0; i < 25; ++i) { w[2 * i] = lo[i]; w[2 * i + 1] = hi[i]; }
    w[50] = pos; w[51] = pos_begin;
  }
  ZK_HD bool fast() const { return pos_begin == 0 && (pos == 32 || pos == 64); }
  // bytes at pos: 00 12 | 40 00 00 00 | (pos + 1) 07 | then the padding (pos + 7) 04 at pos + 8, 80 at byte 167
  ZK_HD scm draw() {
    if (pos == 32) {          // bytes 32..41: words 8, 9, 10  = (lo[4], hi[4], lo[5])
      lo[4] ^= 0x00401200u; hi[4] ^= 0x07210000u; lo[5] ^= 0x00000427u;
    } else {                  // bytes 64..73: words 16, 17, 18 = (lo[8], hi[8], lo[9])
      lo[8] ^= 0x00401200u; hi[8] ^= 0x07410000u; lo[9] ^= 0x00000447u;
    }
    hi[20] ^= 0x80000000u;    // byte 167 = word 41, top byte
#if defined(__HIP_DEVICE_COMPILE__)
    keccak_f1600_halves(lo, hi);
#else
    uint64_t st[25];
    for (int i = 0; i < 25; ++i) st[i] = (uint64_t)lo[i] | ((uint64_t)hi[i] << 32);
    keccak_f1600(st);
    for (int i = 0; i < 25; ++i) { lo[i] = (uint32_t)st[i]; hi[i] = (uint32_t)(st[i] >> 32); }
#endif
    uint32_t wd[16];
    ZK_UNROLL for (int i = 0; i < 8; ++i) { wd[2 * i] = lo[i]; wd[2 * i + 1] = hi[i]; lo[i] = 0; hi[i] = 0; }
    pos = 64;
    pos_begin = 0;
    return scm_from_wide(wd);
  }
};
// The same draws with the generator's state spread over the lanes of a wavefront (keccak_coop.hpp): one wavefront per
// proof, ~55 instructions per Keccak round instead of ~180 dependent ones on a single lane, which shortens the chain of
// 3 + 2 n dependent permutations a commitment phase waits for.  T: the cross-lane traits (device: DevKcTraits, host
// emulation: coop::HostTraits); masks select the lanes (copies included) that hold the state words a draw touches.
template <class T>
struct PvRngCoop {
  using V = typename T::V;
  struct Masks { V w4, w5, w8, w9, w20, keep; };           // holders of words 4, 5, 8, 9, 20; everything but words 0..7
  static ZK_HD_INL void draw(V& lo, V& hi, const typename coop::KeccakCoop<T>::Consts& c, const Masks& m, bool at32) {
    if (at32) {
      lo = T::xor_and(lo, T::splat(0x00401200u), m.w4); hi = T::xor_and(hi, T::splat(0x07210000u), m.w4);
      lo = T::xor_and(lo, T::splat(0x00000427u), m.w5);
    } else {
      lo = T::xor_and(lo, T::splat(0x00401200u), m.w8); hi = T::xor_and(hi, T::splat(0x07410000u), m.w8);
      lo = T::xor_and(lo, T::splat(0x00000447u), m.w9);
    }
    hi = T::xor_and(hi, T::splat(0x80000000u), m.w20);
    coop::KeccakCoop<T>::permute(lo, hi, c);
    // the caller reads words 0..7 (the 64 bytes of the draw) and then calls taken()
  }
  static ZK_HD_INL void taken(V& lo, V& hi, const Masks& m) { lo = T::and_(lo, m.keep); hi = T::and_(hi, m.keep); }
};

#define PV_LBL(s) (s), (unsigned)(sizeof(s) - 1)

// ---- helpers ---------------------------------------------------------------------------------------------
ZK_HD void pv_ld(scm& s, const uint32_t* p) { ZK_UNROLL for (int i = 0; i < 8; ++i) s.v[i] = p[i]; }
ZK_HD void pv_st(uint32_t* p, const scm& s) { ZK_UNROLL for (int i = 0; i < 8; ++i) p[i] = s.v[i]; }
// The building blocks are CALLED, not inlined: a phase is long straight-line code run by one lane per workgroup, and
// with every product and every Keccak-f inlined it outgrows the instruction cache many times over (50 000
// instructions per kernel, and the kernels ran at the speed of instruction fetches).
ZK_HD_NOINLINE scl pv_mul(scl a, scl b) { return scl_mul(a, b); }
// 1 / a mod l by the binary extended Euclidean algorithm on plain 256-bit integers (Stein): ~760 steps of a shift or a
// subtraction on eight words each, against 252 squarings + ~46 products of a^(l-2) -- one lane walks it ~6 times faster.
// Variable time: what is inverted here are Fiat-Shamir challenges (public values).  0 -> 0.
ZK_HD_NOINLINE scm pv_invert(scm am) {
  const uint32_t l[8] = ZK_SC_L;
  uint32_t u[8], v[8], x1[8], x2[8];
  scm_to_words(u, am);
  uint32_t nz = 0;
  ZK_UNROLL for (int i = 0; i < 8; ++i) { v[i] = l[i]; x1[i] = i == 0; x2[i] = 0; nz |= u[i]; }
  if (nz == 0) return scm_zero();
  auto is_one = [](const uint32_t* a) { uint32_t r = a[0] ^ 1u; ZK_UNROLL for (int i = 1; i < 8; ++i) r |= a[i]; return r == 0; };
  auto shr1 = [](uint32_t* a, uint32_t top) {      // (top : a) >> 1
    ZK_UNROLL for (int i = 0; i < 7; ++i) a[i] = (a[i] >> 1) | (a[i + 1] << 31);
    a[7] = (a[7] >> 1) | (top << 31);
  };
  auto halve_mod = [&](uint32_t* x) {              // x / 2 mod l  (x < l)
    uint32_t carry = 0;
    if (x[0] & 1u) {
      uint64_t c = 0;
      ZK_UNROLL for (int i = 0; i < 8; ++i) { c += (uint64_t)x[i] + l[i]; x[i] = (uint32_t)c; c >>= 32; }
      carry = (uint32_t)c;
    }
    shr1(x, carry);
  };
  auto geq = [](const uint32_t* a, const uint32_t* b) {
    bool ge = true, decided = false;
    ZK_UNROLL for (int i = 7; i >= 0; --i) {
      const bool gt = a[i] > b[i], lt = a[i] < b[i];
      ge = decided ? ge : (gt ? true : (lt ? false : ge));
      decided = decided | gt | lt;
    }
    return ge;
  };
  auto sub = [](uint32_t* a, const uint32_t* b) {  // a -= b, returns the borrow
    uint64_t br = 0;
    ZK_UNROLL for (int i = 0; i < 8; ++i) { const uint64_t d = (uint64_t)a[i] - b[i] - br; a[i] = (uint32_t)d; br = (d >> 32) & 1; }
    return (uint32_t)br;
  };
  auto sub_mod = [&](uint32_t* a, const uint32_t* b) {   // a = a - b mod l
    if (sub(a, b)) { uint64_t c = 0; ZK_UNROLL for (int i = 0; i < 8; ++i) { c += (uint64_t)a[i] + l[i]; a[i] = (uint32_t)c; c >>= 32; } }
  };
  ZK_NOUNROLL for (int step = 0; step < 1100; ++step) {
    if (is_one(u) || is_one(v)) break;
    if (!(u[0] & 1u)) { shr1(u, 0); halve_mod(x1); }
    else if (!(v[0] & 1u)) { shr1(v, 0); halve_mod(x2); }
    else if (geq(u, v)) { sub(u, v); sub_mod(x1, x2); }
    else { sub(v, u); sub_mod(x2, x1); }
  }
  return scm_from_words(is_one(u) ? x1 : x2);
}

// The same inverse as a^(l-2) in the lazy form, 253 squarings + 63 products whatever a is: for a kernel that inverts in
// EVERY lane of a wavefront (k_pv_ipa_lanes) -- the Euclidean walk above branches on its data, and 64 lanes walking 64
// different ways cost the wavefront every branch at every step.  0 -> 0.
ZK_HD_NOINLINE scm pv_invert_uniform(scm am) {
  const uint32_t e[8] = ZK_SC_LM2;
  const scl a = scl_from_scm(am);
  scl acc = a;                                     // bit 252 of l - 2
  ZK_NOUNROLL for (int i = 251; i >= 0; --i) {      // (products inlined: two in a loop body, and a call costs a trip to scratch memory)
    acc = scl_mul(acc, acc);
    uint32_t word = 0;
    ZK_UNROLL for (int k = 0; k < 8; ++k) word = (k == (i >> 5)) ? e[k] : word;
    if ((word >> (i & 31)) & 1) acc = scl_mul(acc, a);
  }
  return scl_to_scm(acc);
}
ZK_HD scl pv_ldl(const uint32_t* p) { return scl_from_words(p); }                 // same value, limb form
ZK_HD_NOINLINE void pv_stl(uint32_t* p, scl a) { scl_canon_words(p, a); }         // same value, canonical words
ZK_HD void pv_st_plain(uint32_t* p, const scl& mont) { pv_stl(p, pv_mul(mont, scl_plain_one())); }   // Montgomery -> the integer's words
ZK_HD scl pv_from_plain(const uint32_t* p) { return pv_mul(scl_from_words(p), scl_r2()); }           // canonical integer -> Montgomery

struct PvView {   // the state of one proof
  const PvShape& sh;
  uint32_t* s;
  ZK_HD uint32_t* at(uint32_t off, uint32_t i = 0) const { return s + off + 8 * i; }
};

// value of variable (kind, idx) of the witness, Montgomery limb form
ZK_HD scl pv_var(const PvView& V, uint32_t kind, uint32_t idx) {
  switch (kind) {
    case 0: return pv_ldl(V.at(V.sh.o_v, idx));
    case 1: return pv_ldl(V.at(V.sh.o_aL, idx));
    case 2: return pv_ldl(V.at(V.sh.o_aR, idx));
    case 3: return pv_ldl(V.at(V.sh.o_aO, idx));
    default: return scl_one();
  }
}

// (left, right) of multiplier i from its defining constraints  sum_others coef * var + own * side = 0  (thread 0)
ZK_HD bool pv_solve(const PvView& V, const PvPlan& P, uint32_t con, uint32_t kind, uint32_t i, uint32_t have, scl& out) {
  if (con >= V.sh.n_cons) return false;
  scl acc = scl_zero(), own = scl_zero();
  uint32_t cnt = 0;
  for (uint32_t e = P.con_off[con]; e < P.con_off[con + 1]; ++e) {
    scl c = pv_ldl(P.t_coef + 8 * (uint64_t)e);
    const uint32_t mi = P.t_mono[e];
    if (mi) c = pv_mul(c, pv_ldl(V.at(V.sh.o_sym, mi)));
    const uint32_t tk = P.t_kind[e], ti = P.t_idx[e];
    if (tk == kind && ti == i) { own = scl_weak(scl_add(own, c)); continue; }
    if ((tk >= 1 && tk <= 3 && ti >= have) || (tk == 0 && ti >= V.sh.m)) return false;
    acc = scl_add(acc, tk == 4 ? scl_weak(c) : pv_mul(c, pv_var(V, tk, ti)));
    if (++cnt == 8) { acc = scl_weak(acc); cnt = 1; }
  }
  const scm own_c = scl_to_scm(own), minus_one = scm_neg(scm_one());
  bool zero = true, is_m1 = true;
  for (int q = 0; q < 8; ++q) { zero &= own_c.v[q] == 0; is_m1 &= own_c.v[q] == minus_one.v[q]; }
  if (zero) return false;
  acc = scl_weak(acc);
  if (!is_m1) acc = pv_mul(acc, scl_from_scm(pv_invert(scm_neg(own_c))));
  out = acc;
  return true;
}

// assignments of multipliers [first, last): the given ones by all threads, then the defined ones in index order by thread 0
// (st: 1 the given ones, 2 the defined ones -- see the stages of a phase below)
template <class Env>
ZK_HD void pv_assign(Env& env, const PvView& V, const PvPlan& P, const PvBatch& B, uint32_t proof, uint32_t first, uint32_t last, uint32_t st = 3) {
  const PvShape& sh = V.sh;
  if (st & 1) for (uint32_t i = first + env.tid(); i < last; i += env.nt()) {
    const uint32_t slot = P.given_slot[i];
    if (slot == PV_GIVEN) continue;
    const uint32_t* g = B.given + ((uint64_t)proof * sh.n_given + slot) * 16;
    const scl l = pv_from_plain(g), r = pv_from_plain(g + 8);
    pv_stl(V.at(sh.o_aL, i), l);
    pv_stl(V.at(sh.o_aR, i), r);
    pv_stl(V.at(sh.o_aO, i), pv_mul(l, r));
  }
  env.sync();
  if ((st & 2) && env.tid() == 0) {
    for (uint32_t i = first; i < last; ++i) {
      if (P.given_slot[i] != PV_GIVEN) continue;
      scl l = scl_zero(), r = scl_zero();
      const bool ok = pv_solve(V, P, P.mult_def[2 * i], 1, i, i, l) && pv_solve(V, P, P.mult_def[2 * i + 1], 2, i, i, r);
      if (!ok) V.s[sh.o_flag] = 1;
      pv_stl(V.at(sh.o_aL, i), l);
      pv_stl(V.at(sh.o_aR, i), r);
      pv_stl(V.at(sh.o_aO, i), pv_mul(l, r));
    }
  }
  env.sync();
}

// rows of a Pedersen vector commitment phase over multipliers [first, last): A_I = [i_bl | aL.. | aR..],
// A_O = [o_bl | aO..], S = [s_bl | sL.. | sR..]  as plain canonical words.  The witness part (all threads):
template <class Env>
ZK_HD void pv_commit_rows(Env& env, const PvView& V, uint32_t* rows, uint32_t first, uint32_t last) {
  const PvShape& sh = V.sh;
  const uint32_t cnt = last - first;
  uint32_t* rI = rows;
  uint32_t* rO = rI + 8 * (1 + 2 * cnt);
  for (uint32_t j = env.tid(); j < cnt; j += env.nt()) {
    const uint32_t i = first + j;
    pv_st_plain(rI + 8 * (1 + j), pv_ldl(V.at(sh.o_aL, i)));
    pv_st_plain(rI + 8 * (1 + cnt + j), pv_ldl(V.at(sh.o_aR, i)));
    pv_st_plain(rO + 8 * (1 + j), pv_ldl(V.at(sh.o_aO, i)));
  }
}
// ... and the part drawn from the TranscriptRng, in the reference's order: i_bl, o_bl, s_bl, sL.., sR.. (ONE thread per
// proof: k_pv_rng runs it one lane per proof).  Values go to the state (Montgomery) and to the rows (plain).
ZK_HD void pv_rng_draw(const PvShape& sh, uint32_t* state, uint32_t* rows, uint32_t first, uint32_t last, uint32_t bl_slot) {
  PvView V{sh, state};
  const uint32_t cnt = last - first;
  uint32_t* rI = rows;
  uint32_t* rO = rI + 8 * (1 + 2 * cnt);
  uint32_t* rS = rO + 8 * (1 + cnt);
  PvRng rng;
  rng.load(state + sh.o_rng);
  if (!rng.fast()) { state[sh.o_flag] = 2; return; }   // cannot happen: the generator was keyed just before
  auto one = [&](uint32_t* st_slot, uint32_t* row_slot) {
    const scm v = rng.draw();
    pv_st(st_slot, v);
    pv_st_plain(row_slot, scl_from_scm(v));
  };
  one(V.at(sh.o_blind, bl_slot), rI);
  one(V.at(sh.o_blind, bl_slot + 1), rO);
  one(V.at(sh.o_blind, bl_slot + 2), rS);
  for (uint32_t j = 0; j < cnt; ++j) one(V.at(sh.o_sL, first + j), rS + 8 * (1 + j));
  for (uint32_t j = 0; j < cnt; ++j) one(V.at(sh.o_sR, first + j), rS + 8 * (1 + cnt + j));
  rng.store(state + sh.o_rng);
}

// ---- phase 0: the value commitments' rows (no transcript yet) ----------------------------------------------
template <class Env>
ZK_HD void pv_phase0(Env& env, const PvShape& sh, const PvBatch& B, uint32_t proof) {
  PvView V{sh, B.state + (uint64_t)proof * sh.state_words};
  for (uint32_t j = env.tid(); j < sh.m; j += env.nt()) {
    const uint32_t* v = B.values + ((uint64_t)proof * sh.m + j) * 8;
    const uint32_t* bl = B.blindings + ((uint64_t)proof * sh.m + j) * 8;
    uint32_t* row = B.rows0 + ((uint64_t)proof * sh.m + j) * 16;
    const scl vm = pv_from_plain(v), bm = pv_from_plain(bl);      // any 256-bit input, reduced mod l
    pv_stl(V.at(sh.o_v, j), vm);
    pv_stl(V.at(sh.o_vbl, j), bm);
    pv_st_plain(row, vm);
    pv_st_plain(row + 8, bm);
  }
  if (env.tid() == 0) V.s[sh.o_flag] = 0;
}

// ---- phase 1: V_j in; first-phase witness, blinding vectors; rows of A_I1 A_O1 S1 ----------------------------
// STAGES of a phase (st, a mask; PV_ALL = the whole phase in one go, as the host runs it): a phase alternates between work
// of ONE thread per proof (transcript, challenges, the multipliers defined by constraints, the draws) and work of the whole
// workgroup.  On the device every stage is a launch of its own: the one-thread stages with one LANE per proof (64 proofs to a
// wavefront: k_pv_lanes), the others with a workgroup per proof (k_pv_wg) -- inside one kernel the one-thread stages kept one
// lane of one wavefront busy while the rest of the workgroup waited: 6 % of a call's instructions on 1/64 of the lanes, and
// the longest part of a workgroup's life.   phase 1, 2: 1 transcript | 2 given multipliers | 4 defined multipliers | 8 rows
//                                            phase 3: 1 transcript, y z 1/y | 2 tables, flattening, t_i | 4 draws, rows
//                                            phase 4: 1 transcript .. w | 2 l(x), r(x), generator coefficients
constexpr uint32_t PV_ALL = 15;
template <class Env>
ZK_HD void pv_phase1(Env& env, const PvShape& sh, const PvPlan& P, const PvBatch& B, uint32_t proof, const uint32_t* v_points /*m x 8*/, uint32_t st = PV_ALL) {
  PvView V{sh, B.state + (uint64_t)proof * sh.state_words};
  if ((st & 1) && env.tid() == 0) {
    PvStrobe tr{env.strobe()};
    for (int i = 0; i < 52; ++i) tr.w[i] = P.init[i];
    for (uint32_t j = 0; j < sh.m; ++j) tr.append_words(PV_LBL("V"), v_points + 8 * j, 8);
    tr.append_u64(PV_LBL("m"), sh.m);
    for (int i = 0; i < 52; ++i) V.s[sh.o_tr + i] = tr.w[i];
    // TranscriptRng: a fork of the transcript, keyed with the blinding factors and the external randomness
    for (uint32_t j = 0; j < sh.m; ++j) tr.rekey_with_witness(PV_LBL("v_blinding"), B.rows0 + ((uint64_t)proof * sh.m + j) * 16 + 8);
    tr.finalize_rng(B.rng_seed + (uint64_t)proof * 8);
    for (int i = 0; i < 52; ++i) V.s[sh.o_rng + i] = tr.w[i];   // the draws follow in pv_rng_draw
    pv_st(V.at(sh.o_sym, 0), scm_one());
  }
  env.sync();
  pv_assign(env, V, P, B, proof, 0, sh.n1, (st >> 1) & 3);
  if (st & 8) pv_commit_rows(env, V, B.rows1 + (uint64_t)proof * sh.r1_terms * 8, 0, sh.n1);
}

// ---- phase 2: A_I1 A_O1 S1 in; second-phase challenges and witness; rows of A_I2 A_O2 S2 ----------------------
template <class Env>
ZK_HD void pv_phase2(Env& env, const PvShape& sh, const PvPlan& P, const PvBatch& B, uint32_t proof, const uint32_t* pts /*3 x 8*/, uint32_t st = PV_ALL) {
  PvView V{sh, B.state + (uint64_t)proof * sh.state_words};
  uint8_t* proof_bytes = B.proofs + (uint64_t)proof * sh.proof_stride;
  if ((st & 1) && env.tid() == 0) {
    PvStrobe tr{env.strobe()};
    for (int i = 0; i < 52; ++i) tr.w[i] = V.s[sh.o_tr + i];
    tr.append_words(PV_LBL("A_I1"), pts, 8);
    tr.append_words(PV_LBL("A_O1"), pts + 8, 8);
    tr.append_words(PV_LBL("S1"), pts + 16, 8);
    proof_bytes[0] = 1;   // two-phase wire format
    for (int q = 0; q < 24; ++q) for (int b = 0; b < 4; ++b) proof_bytes[1 + 4 * q + b] = (uint8_t)(pts[q] >> (8 * b));
    if (!sh.two_phase) {
      tr.append_message(PV_LBL("dom-sep"), (const uint8_t*)"r1cs-1phase", 11);
    } else {
      tr.append_message(PV_LBL("dom-sep"), (const uint8_t*)"r1cs-2phase", 11);
      for (uint32_t j = 0; j < sh.n_chal2; ++j) {
        const uint8_t* lab = P.chal_labels + 32 * j;
        pv_st(V.at(sh.o_c2, j), tr.challenge_scalar((const char*)(lab + 1), lab[0]));
      }
      for (uint32_t mi = 1; mi < sh.n_mono; ++mi) {   // monomials c^p of the second-phase challenges
        const scl base = pv_ldl(V.at(sh.o_c2, P.mono_chal[mi]));
        scl acc = base;
        for (uint32_t e = 1; e < P.mono_pow[mi]; ++e) acc = pv_mul(acc, base);
        pv_stl(V.at(sh.o_sym, mi), acc);
      }
    }
    for (int i = 0; i < 52; ++i) V.s[sh.o_tr + i] = tr.w[i];
  }
  env.sync();
  const uint32_t n2 = sh.n - sh.n1;
  if (n2 == 0) {                      // three empty rows (the identity), no second-phase blinding factors
    if ((st & 1) && env.tid() == 0) for (int k = PV_IBL2; k <= PV_SBL2; ++k) pv_st(V.at(sh.o_blind, k), scm_zero());
    return;
  }
  pv_assign(env, V, P, B, proof, sh.n1, sh.n, (st >> 1) & 3);
  if (st & 8) pv_commit_rows(env, V, B.rows2 + (uint64_t)proof * sh.r2_terms * 8, sh.n1, sh.n);   // the drawn part: pv_rng_draw
}

// ---- phase 3: A_I2 A_O2 S2 in; y, z; flattening, t(x) coefficients; rows of T_1 T_3 T_4 T_5 T_6 ----------------
// l(x) = l1 x + l2 x^2 + l3 x^3, r(x) = r0 + r1 x + r3 x^3 with
//   l1 = aL + y^-i wR, l2 = aO, l3 = sL, r0 = wO - y^i, r1 = y^i aR + wL, r3 = y^i sR
struct PvPoly { scl l1, l2, l3, r0, r1, r3; };
ZK_HD PvPoly pv_poly(const PvView& V, uint32_t i) {
  const PvShape& sh = V.sh;
  const scl yp = pv_ldl(V.at(sh.o_ypow, i)), yi = pv_ldl(V.at(sh.o_yinv, i));
  PvPoly p;
  p.l1 = scl_add(pv_ldl(V.at(sh.o_aL, i)), pv_mul(yi, pv_ldl(V.at(sh.o_wR, i))));
  p.l2 = pv_ldl(V.at(sh.o_aO, i));
  p.l3 = pv_ldl(V.at(sh.o_sL, i));
  p.r0 = scl_sub(pv_ldl(V.at(sh.o_wO, i)), yp);
  p.r1 = scl_add(pv_mul(yp, pv_ldl(V.at(sh.o_aR, i))), pv_ldl(V.at(sh.o_wL, i)));
  p.r3 = pv_mul(yp, pv_ldl(V.at(sh.o_sR, i)));
  return p;
}

// table[i] = base^i for i < count, by doubling: entries [half, 2 half) = entries [0, half) * base^half
template <class Env>
ZK_HD void pv_powers(Env& env, uint32_t* table, const scl& base, uint32_t count) {
  if (env.tid() == 0) pv_stl(table, scl_one());
  env.sync();
  scl stride = base;
  for (uint32_t half = 1; half < count; half <<= 1) {
    const uint32_t end = 2 * half < count ? 2 * half : count;
    for (uint32_t q = half + env.tid(); q < end; q += env.nt()) pv_stl(table + 8 * q, pv_mul(pv_ldl(table + 8 * (q - half)), stride));
    env.sync();
    stride = pv_mul(stride, stride);
  }
}

template <class Env>
ZK_HD void pv_phase3(Env& env, const PvShape& sh, const PvPlan& P, const PvBatch& B, uint32_t proof, const uint32_t* pts /*3 x 8*/, uint32_t st = PV_ALL) {
  PvView V{sh, B.state + (uint64_t)proof * sh.state_words};
  uint8_t* proof_bytes = B.proofs + (uint64_t)proof * sh.proof_stride;
  if ((st & 1) && env.tid() == 0) {
    PvStrobe tr{env.strobe()};
    for (int i = 0; i < 52; ++i) tr.w[i] = V.s[sh.o_tr + i];
    tr.append_words(PV_LBL("A_I2"), pts, 8);
    tr.append_words(PV_LBL("A_O2"), pts + 8, 8);
    tr.append_words(PV_LBL("S2"), pts + 16, 8);
    for (int q = 0; q < 24; ++q) for (int b = 0; b < 4; ++b) proof_bytes[1 + 96 + 4 * q + b] = (uint8_t)(pts[q] >> (8 * b));
    const scm y = tr.challenge_scalar(PV_LBL("y")), z = tr.challenge_scalar(PV_LBL("z"));
    pv_st(V.at(sh.o_chal, PV_Y), y);
    pv_st(V.at(sh.o_chal, PV_Z), z);
    pv_st(V.at(sh.o_chal, PV_YINV), Env::kInvertInEveryLane ? pv_invert_uniform(y) : pv_invert(y));
    for (int i = 0; i < 52; ++i) V.s[sh.o_tr + i] = tr.w[i];
  }
  env.sync();
  if (st & 2) {
  const scl y = pv_ldl(V.at(sh.o_chal, PV_Y)), z = pv_ldl(V.at(sh.o_chal, PV_Z)), yinv = pv_ldl(V.at(sh.o_chal, PV_YINV));
  // z^(q+1) for q < n_cons: table of z^i shifted by one
  pv_powers(env, V.at(sh.o_zpow), z, sh.n_cons + 1);
  pv_powers(env, V.at(sh.o_ypow), y, sh.pn);
  pv_powers(env, V.at(sh.o_yinv), yinv, sh.pn);
  // flattening: wL | wR | wO | wV (wV negated, as the verifier's plan has it: sign folded into the terms)
  const uint32_t n_tgt = 3 * sh.n + sh.m;
  for (uint32_t g = env.tid(); g < n_tgt; g += env.nt()) {
    scl acc = scl_zero();
    uint32_t cnt = 0;
    for (uint32_t e = P.tgt_off[g]; e < P.tgt_off[g + 1]; ++e) {
      const uint32_t info = P.term_info[e];
      scl v;
      if (info & 0x80000000u) {
        v = pv_ldl(V.at(sh.o_zpow, (info & 0x00ffffffu) + 1));
      } else {
        const uint32_t p = info & 0x00ffffffu, q = P.prod_qm[2 * p], mi = P.prod_qm[2 * p + 1];
        scl c;
        for (int k = 0; k < 10; ++k) c.v[k] = P.prod_coef[10 * (uint64_t)p + k];
        if (mi) c = pv_mul(c, pv_ldl(V.at(sh.o_sym, mi)));
        v = pv_mul(c, pv_ldl(V.at(sh.o_zpow, q + 1)));
      }
      acc = scl_add(acc, scl_cneg(v, (info & 0x40000000u) != 0));
      if (++cnt == 16) { acc = scl_weak(acc); cnt = 1; }
    }
    uint32_t* dst = g < sh.n ? V.at(sh.o_wL, g) : g < 2 * sh.n ? V.at(sh.o_wR, g - sh.n) : g < 3 * sh.n ? V.at(sh.o_wO, g - 2 * sh.n) : V.at(sh.o_wV, g - 3 * sh.n);
    pv_stl(dst, acc);
  }
  env.sync();
  // t_1 .. t_6
  scl t[6];
  for (int k = 0; k < 6; ++k) t[k] = scl_zero();
  uint32_t cnt = 0;
  for (uint32_t i = env.tid(); i < sh.n; i += env.nt()) {
    const PvPoly p = pv_poly(V, i);
    // operands of a product: limbs < 2^28 each (sums of two tight values or a difference)
    t[0] = scl_add(t[0], pv_mul(p.l1, p.r0));
    t[1] = scl_add(t[1], scl_add(pv_mul(p.l1, p.r1), pv_mul(p.l2, p.r0)));
    t[2] = scl_add(t[2], scl_add(pv_mul(p.l2, p.r1), pv_mul(p.l3, p.r0)));
    t[3] = scl_add(t[3], scl_add(pv_mul(p.l1, p.r3), pv_mul(p.l3, p.r1)));
    t[4] = scl_add(t[4], pv_mul(p.l2, p.r3));
    t[5] = scl_add(t[5], pv_mul(p.l3, p.r3));
    if (++cnt == 8) { for (int k = 0; k < 6; ++k) t[k] = scl_weak(t[k]); cnt = 1; }
  }
  for (int k = 0; k < 6; ++k) t[k] = scl_weak(t[k]);
  env.sum(t, 6);                      // thread 0 holds the sums
  if (env.tid() == 0) for (int k = 0; k < 6; ++k) pv_stl(V.at(sh.o_t, k + 1), t[k]);
  }
  if ((st & 4) && env.tid() == 0) {
    PvRng rng;
    rng.load(V.s + sh.o_rng);
    if (!rng.fast()) V.s[sh.o_flag] = 2;
    const int order[5] = {1, 3, 4, 5, 6};
    pv_st(V.at(sh.o_tb, 2), scm_zero());
    for (int k = 0; k < 5; ++k) pv_st(V.at(sh.o_tb, order[k]), rng.draw());
    rng.store(V.s + sh.o_rng);
    uint32_t* rows = B.rows3 + (uint64_t)proof * 5 * 16;
    for (int k = 0; k < 5; ++k) {
      pv_st_plain(rows + 16 * k, pv_ldl(V.at(sh.o_t, order[k])));
      pv_st_plain(rows + 16 * k + 8, pv_ldl(V.at(sh.o_tb, order[k])));
    }
  }
}

// ---- phase 4: T points in; u, x; t(x), blindings, l(x), r(x); the start of the inner-product argument ------------
template <class Env>
ZK_HD void pv_phase4(Env& env, const PvShape& sh, const PvPlan& P, const PvBatch& B, uint32_t proof, const uint32_t* pts /*5 x 8*/, uint32_t st = PV_ALL) {
  PvView V{sh, B.state + (uint64_t)proof * sh.state_words};
  uint8_t* proof_bytes = B.proofs + (uint64_t)proof * sh.proof_stride;
  (void)P;
  if ((st & 1) && env.tid() == 0) {
    PvStrobe tr{env.strobe()};
    for (int i = 0; i < 52; ++i) tr.w[i] = V.s[sh.o_tr + i];
    tr.append_words(PV_LBL("T_1"), pts, 8);
    tr.append_words(PV_LBL("T_3"), pts + 8, 8);
    tr.append_words(PV_LBL("T_4"), pts + 16, 8);
    tr.append_words(PV_LBL("T_5"), pts + 24, 8);
    tr.append_words(PV_LBL("T_6"), pts + 32, 8);
    for (int q = 0; q < 40; ++q) for (int b = 0; b < 4; ++b) proof_bytes[1 + 192 + 4 * q + b] = (uint8_t)(pts[q] >> (8 * b));
    const scm um = tr.challenge_scalar(PV_LBL("u")), xm = tr.challenge_scalar(PV_LBL("x"));
    pv_st(V.at(sh.o_chal, PV_U), um);
    pv_st(V.at(sh.o_chal, PV_X), xm);
    const scl u = scl_from_scm(um), x = scl_from_scm(xm);
    scl tb2 = scl_zero();
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < sh.m; ++j) {
      tb2 = scl_add(tb2, pv_mul(pv_ldl(V.at(sh.o_wV, j)), pv_ldl(V.at(sh.o_vbl, j))));
      if (++cnt == 16) { tb2 = scl_weak(tb2); cnt = 1; }
    }
    tb2 = scl_weak(tb2);
    pv_stl(V.at(sh.o_tb, 2), tb2);
    scl xp = scl_one(), t_x = scl_zero(), t_x_bl = scl_zero();
    for (int i = 1; i <= 6; ++i) {
      xp = pv_mul(xp, x);
      t_x = scl_add(t_x, pv_mul(pv_ldl(V.at(sh.o_t, i)), xp));
      t_x_bl = scl_add(t_x_bl, pv_mul(pv_ldl(V.at(sh.o_tb, i)), xp));
    }
    const scl i_bl = scl_add(pv_ldl(V.at(sh.o_blind, PV_IBL1)), pv_mul(u, pv_ldl(V.at(sh.o_blind, PV_IBL2))));
    const scl o_bl = scl_add(pv_ldl(V.at(sh.o_blind, PV_OBL1)), pv_mul(u, pv_ldl(V.at(sh.o_blind, PV_OBL2))));
    const scl s_bl = scl_add(pv_ldl(V.at(sh.o_blind, PV_SBL1)), pv_mul(u, pv_ldl(V.at(sh.o_blind, PV_SBL2))));
    scl e_bl = pv_mul(scl_add(pv_mul(x, s_bl), o_bl), x);
    e_bl = pv_mul(scl_add(e_bl, i_bl), x);
    uint32_t w3[24];
    pv_st_plain(w3, t_x);
    pv_st_plain(w3 + 8, t_x_bl);
    pv_st_plain(w3 + 16, e_bl);
    tr.append_words(PV_LBL("t_x"), w3, 8);
    tr.append_words(PV_LBL("t_x_blinding"), w3 + 8, 8);
    tr.append_words(PV_LBL("e_blinding"), w3 + 16, 8);
    for (int q = 0; q < 24; ++q) for (int b = 0; b < 4; ++b) proof_bytes[1 + 352 + 4 * q + b] = (uint8_t)(w3[q] >> (8 * b));
    const scm w = tr.challenge_scalar(PV_LBL("w"));
    pv_st(V.at(sh.o_chal, PV_W), w);
    pv_st_plain(B.ipa_w + (uint64_t)proof * 8, scl_from_scm(w));
    tr.append_message(PV_LBL("dom-sep"), (const uint8_t*)"ipp v1", 6);
    tr.append_u64(PV_LBL("n"), sh.pn);
    for (int i = 0; i < 52; ++i) V.s[sh.o_tr + i] = tr.w[i];
  }
  env.sync();
  if (!(st & 2)) return;
  const scl x = pv_ldl(V.at(sh.o_chal, PV_X)), u = pv_ldl(V.at(sh.o_chal, PV_U));
  const scl x2 = pv_mul(x, x), x3 = pv_mul(x2, x);
  uint32_t* lv = B.ipa_lv + (uint64_t)proof * sh.pn * 8;
  uint32_t* rv = B.ipa_rv + (uint64_t)proof * sh.pn * 8;
  uint32_t* cg = B.ipa_cg + (uint64_t)proof * sh.pn * 8;
  uint32_t* ch = B.ipa_ch + (uint64_t)proof * sh.pn * 8;
  for (uint32_t i = env.tid(); i < sh.pn; i += env.nt()) {
    scl l = scl_zero(), r;
    if (i < sh.n) {
      const PvPoly p = pv_poly(V, i);
      l = scl_add(scl_add(pv_mul(p.l1, x), pv_mul(p.l2, x2)), pv_mul(p.l3, x3));
      r = scl_add(scl_add(p.r0, pv_mul(p.r1, x)), pv_mul(p.r3, x3));
    } else {
      r = scl_neg(pv_ldl(V.at(sh.o_ypow, i)));
    }
    pv_stl(lv + 8 * i, l);              // Montgomery form, as k_ipa_round keeps them
    pv_stl(rv + 8 * i, r);
    const scl g = i < sh.n1 ? scl_one() : u;
    pv_st_plain(cg + 8 * i, g);         // generator coefficients as plain words
    pv_st_plain(ch + 8 * i, pv_mul(pv_ldl(V.at(sh.o_yinv, i)), g));
  }
}

// ---- one round of the inner-product argument: L_j, R_j in; the round's challenge and its inverse out -----------
template <class Env>
ZK_HD void pv_ipa_round(Env& env, const PvShape& sh, const PvBatch& B, uint32_t proof, uint32_t round, const uint32_t* lr /*2 x 8*/) {
  PvView V{sh, B.state + (uint64_t)proof * sh.state_words};
  uint8_t* proof_bytes = B.proofs + (uint64_t)proof * sh.proof_stride;
  if (env.tid() != 0) return;
  PvStrobe tr{env.strobe()};
  for (int i = 0; i < 52; ++i) tr.w[i] = V.s[sh.o_tr + i];
  tr.append_words(PV_LBL("L"), lr, 8);
  tr.append_words(PV_LBL("R"), lr + 8, 8);
  for (int q = 0; q < 16; ++q) for (int b = 0; b < 4; ++b) proof_bytes[1 + 448 + 64 * round + 4 * q + b] = (uint8_t)(lr[q] >> (8 * b));
  const scm u = tr.challenge_scalar(PV_LBL("u"));
  uint32_t* out = B.ipa_u + (uint64_t)proof * 16;
  pv_st_plain(out, scl_from_scm(u));
  pv_st_plain(out + 8, scl_from_scm(Env::kInvertInEveryLane ? pv_invert_uniform(u) : pv_invert(u)));
  for (int i = 0; i < 52; ++i) V.s[sh.o_tr + i] = tr.w[i];
}

// a, b (canonical words from k_ipa_round) -> the proof's tail
ZK_HD void pv_finish(const PvShape& sh, const PvBatch& B, uint32_t proof, const uint32_t* ab /*2 x 8*/) {
  uint8_t* proof_bytes = B.proofs + (uint64_t)proof * sh.proof_stride;
  for (int q = 0; q < 16; ++q) for (int b = 0; b < 4; ++b) proof_bytes[1 + 448 + 64 * sh.k + 4 * q + b] = (uint8_t)(ab[q] >> (8 * b));
}

}  // namespace zk
