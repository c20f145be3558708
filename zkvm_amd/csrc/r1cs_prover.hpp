// r1cs_prover.hpp -- host side of bulletproofs::r1cs::Prover for ZkVM `cloak`
// statements, written as a sequence of PHASES: every phase hands out the
// multiscalar multiplications it needs as rows over the resident generator set
// [B, B_blinding, G_0.., H_0..] and the next phase absorbs their encodings.
// A batch of provers runs in lockstep, so that each phase of the whole batch
// is ONE zkgpu_msm_ps_batch call on the fixed-base tables (SURVEY.md sec 8 row
// f-4, BASELINE.json configs[4]; upstream `r1cs::Prover::prove`,
// `InnerProductProof::create`, `spacesuit::cloak` -- sources not mounted; the
// byte-level behaviour follows oracle/r1cs.c / oracle/cloak.c, which restate the
// same recollection, and the CPU tests require byte-identical proofs).
//
// The inner-product argument keeps the folded generators as coefficient vectors
// over the ORIGINAL generators, so every L_j / R_j is a multiscalar
// multiplication over the resident set and the tables apply to all of them.
//
// The gadget code is the verifier's (r1cs_verifier.hpp, namespace cloak): the
// prover is a ConstraintSystemT<Scalar> whose multiply() evaluates the linear
// combinations on the witness and whose allocate_multiplier() takes the
// assignments of first-phase allocations from a queue that cloak_witness()
// fills in the gadget's allocation order.
#pragma once
#include "r1cs_verifier.hpp"
#include "cloak_plan.hpp"

#include <algorithm>
#include <deque>
#include <functional>
#include <memory>

namespace zk {

struct MsmRow {
  std::vector<Scalar> scalars;
  std::vector<uint32_t> index;   // into the generator set: 0 B, 1 B_blinding, 2 + i G_i, 2 + cap + i H_i
  void add(const Scalar& s, uint32_t i) { scalars.push_back(s); index.push_back(i); }
};

class R1csProverCS : public ConstraintSystemT<Scalar> {
 public:
  explicit R1csProverCS(const char* label) : tr(label) { tr.append_message("dom-sep", (const uint8_t*)"r1cs v1", 7); }

  Var commit(const Scalar& value, const Scalar& blinding) {
    v.push_back(value);
    v_bl.push_back(blinding);
    return Var{VarKind::Committed, (uint32_t)(v.size() - 1)};
  }
  Scalar challenge_scalar(const char* label) override { return tr.challenge_scalar(label); }

  Scalar eval(const LC& lc) const {
    Scalar acc = Scalar::zero();
    for (const auto& t : lc.terms) {
      switch (t.first.kind) {
        case VarKind::MulLeft: acc += aL[t.first.idx] * t.second; break;
        case VarKind::MulRight: acc += aR[t.first.idx] * t.second; break;
        case VarKind::MulOut: acc += aO[t.first.idx] * t.second; break;
        case VarKind::Committed: acc += v[t.first.idx] * t.second; break;
        case VarKind::One: acc += t.second; break;
      }
    }
    return acc;
  }
  void multiply(LC left, LC right, Var out[3]) override {
    pending_l_ = eval(left);
    pending_r_ = eval(right);
    has_pending_ = true;
    ConstraintSystemT<Scalar>::multiply(std::move(left), std::move(right), out);
  }
  void allocate_multiplier(Var out[3]) override {
    Scalar l, r;
    if (has_pending_) {
      l = pending_l_; r = pending_r_;
      has_pending_ = false;
    } else {
      if (queue.empty()) { failed = true; l = r = Scalar::zero(); }
      else { l = queue.front().first; r = queue.front().second; queue.pop_front(); }
    }
    aL.push_back(l); aR.push_back(r); aO.push_back(l * r);
    ConstraintSystemT<Scalar>::allocate_multiplier(out);
  }

  // protected members of the base, for the prover proper
  size_t second_phase() { return run_second_phase(); }
  bool deferred() const { return has_deferred(); }
  const std::vector<LC>& constraints() const { return cons_; }
  size_t n_vars() const { return num_vars_; }

  Transcript tr;
  std::vector<Scalar> aL, aR, aO, v, v_bl;
  std::deque<std::pair<Scalar, Scalar>> queue;   // assignments of first-phase allocate_multiplier calls, in order
  bool failed = false;

 private:
  Scalar pending_l_, pending_r_;
  bool has_pending_ = false;
};

namespace cloak {

struct Amount {   // one value of the witness
  uint64_t q;
  Scalar f;
};

inline bool flavor_less(const Scalar& a, const Scalar& b) {   // numeric order of the canonical values
  for (int i = 3; i >= 0; --i) if (a.v[i] != b.v[i]) return a.v[i] < b.v[i];
  return false;
}

// assignments of one k_mix in its allocation order: grouped (k), mid (k - 2), merged (k)
inline void k_mix_witness(const std::vector<Amount>& vals, std::deque<std::pair<Scalar, Scalar>>& out,
                          std::vector<Amount>* merged_out) {
  const size_t k = vals.size();
  if (merged_out) *merged_out = vals;
  if (k <= 1) return;
  std::vector<Amount> g = vals;   // stable insertion sort by flavor
  for (size_t i = 1; i < k; ++i) {
    const Amount a = g[i];
    size_t j = i;
    while (j > 0 && flavor_less(a.f, g[j - 1].f)) { g[j] = g[j - 1]; --j; }
    g[j] = a;
  }
  std::vector<Amount> m(k), d(k);
  Amount acc = g[0];
  for (size_t i = 1; i < k; ++i) {
    if (acc.f == g[i].f) { m[i - 1] = Amount{0, Scalar::zero()}; acc.q += g[i].q; }
    else { m[i - 1] = acc; acc = g[i]; }
    d[i - 1] = acc;   // running D after step i
  }
  m[k - 1] = acc;
  auto push = [&](const Amount& a) { out.emplace_back(Scalar::from_u64(a.q), a.f); };
  for (size_t i = 0; i < k; ++i) push(g[i]);
  for (size_t i = 0; i + 2 < k; ++i) push(d[i]);
  for (size_t i = 0; i < k; ++i) push(m[i]);
  if (merged_out) *merged_out = m;
}

// every first-phase allocation of cloak::gadget, in order (k_mix in, k_mix out, shuffle padding, range bits)
inline void gadget_witness(const std::vector<Amount>& in, const std::vector<Amount>& out,
                           std::deque<std::pair<Scalar, Scalar>>& q) {
  k_mix_witness(in, q, nullptr);
  k_mix_witness(out, q, nullptr);
  const size_t pad = in.size() > out.size() ? in.size() - out.size() : out.size() - in.size();
  for (size_t i = 0; i < pad; ++i) q.emplace_back(Scalar::zero(), Scalar::zero());
  for (const Amount& o : out)
    for (int i = 0; i < 64; ++i) {
      const uint64_t bit = (o.q >> i) & 1;
      q.emplace_back(Scalar::from_u64(1 - bit), Scalar::from_u64(bit));
    }
}

}  // namespace cloak

// One proof in the making.  Usage: begin(rows); then alternately evaluate the rows (one encoding
// per row, in order) and call step(points, rows) until done(); then proof() / commitments().
//
// The statement comes in as (committed values, their blindings, a SYNTHESIS function): the function is
// called once, after the commitments went into the transcript, with the constraint system and the
// committed variables; it supplies the first-phase witness (cs.queue) and runs the gadget code --
// cloak_prover() below for the ZkVM cloak, desc_prover() for a constraint system described as data.
class R1csProver {
 public:
  using Synthesis = std::function<void(R1csProverCS&, const std::vector<Var>&)>;

  // seed: 32 bytes from which the external randomness of the TranscriptRng is derived
  // (SHAKE256(seed || "rng" || LE64(0)), as the oracle's provers)
  R1csProver(const char* label, const uint8_t seed[32], size_t gens_capacity, std::vector<Scalar> values,
             std::vector<Scalar> blindings, Synthesis synth)
      : cs_(label), cap_(gens_capacity), values_(std::move(values)), blindings_(std::move(blindings)), synth_(std::move(synth)) {
    std::memcpy(seed_, seed, 32);
  }

  static Scalar derive_scalar(const uint8_t seed[32], const char* tag, uint64_t i) {
    uint8_t wide[64];
    derive(seed, tag, i, wide, 64);
    return Scalar::from_wide(wide);
  }
  static void derive(const uint8_t seed[32], const char* tag, uint64_t i, uint8_t* out, size_t n) {
    Sponge sp = shake256_sponge();
    uint8_t ib[8];
    for (int k = 0; k < 8; ++k) ib[k] = (uint8_t)(i >> (8 * k));
    sp.absorb(seed, 32);
    sp.absorb((const uint8_t*)tag, std::strlen(tag));
    sp.absorb(ib, 8);
    sp.squeeze(out, n);
  }

  bool done() const { return stage_ == kDone; }
  // ---- the inner-product argument on the device (zkgpu.hip, ipa_on_device): the vectors leave after the
  // polynomial phase, the prover keeps only its transcript going ----
  void set_device_ipa(bool on) { device_ipa_ = on; }
  bool at_ipa() const { return stage_ == kIpa && device_ipa_; }
  size_t ipa_len() const { return pn_; }
  size_t ipa_rounds() const { return k_; }
  size_t gens_capacity() const { return cap_; }
  // canonical 32-byte scalars: l, r, the generator coefficient vectors (pn each) and the weight w of Q = w B
  void ipa_export(uint8_t* lv, uint8_t* rv, uint8_t* cg, uint8_t* ch, uint8_t* w) const {
    for (size_t i = 0; i < pn_; ++i) { lv_[i].to_bytes(lv + 32 * i); rv_[i].to_bytes(rv + 32 * i); cG_[i].to_bytes(cg + 32 * i); cH_[i].to_bytes(ch + 32 * i); }
    w_.to_bytes(w);
  }
  // L_j, R_j of the current round in; the round's challenge u and 1/u out (canonical)
  void ipa_absorb(const uint8_t lr[64], uint8_t u_and_inverse[64]) {
    proof_.insert(proof_.end(), lr, lr + 64);
    cs_.tr.append_point("L", lr);
    cs_.tr.append_point("R", lr + 32);
    const Scalar uu = cs_.tr.challenge_scalar("u"), uu_inv = uu.invert();
    uu.to_bytes(u_and_inverse);
    uu_inv.to_bytes(u_and_inverse + 32);
  }
  void ipa_finish(const uint8_t a[32], const uint8_t b[32]) {
    proof_.insert(proof_.end(), a, a + 32);
    proof_.insert(proof_.end(), b, b + 32);
    stage_ = kDone;
  }
  bool failed() const { return failed_; }
  const std::vector<uint8_t>& proof() const { return proof_; }
  const std::vector<uint8_t>& commitments() const { return commitments_; }
  size_t multipliers() const { return cs_.n_vars(); }

  void begin(std::vector<MsmRow>& rows) {
    rows.clear();
    for (size_t i = 0; i < values_.size(); ++i) {
      vars_.push_back(cs_.commit(values_[i], blindings_[i]));
      MsmRow r;
      r.add(values_[i], 0); r.add(blindings_[i], 1);
      rows.push_back(std::move(r));
    }
    stage_ = kCommitted;
  }

  // points: 32 bytes per row of the previous call, in order
  void step(const uint8_t* points, std::vector<MsmRow>& rows) {
    rows.clear();
    switch (stage_) {
      case kCommitted: after_commitments(points, rows); break;
      case kPhase1: after_phase1(points, rows); break;
      case kPhase2: after_phase2(points, rows); break;
      case kT: after_t(points, rows); break;
      case kIpa: after_ipa_round(points, rows); break;
      default: failed_ = true; stage_ = kDone; break;
    }
  }

 private:
  enum Stage { kNew, kCommitted, kPhase1, kPhase2, kT, kIpa, kDone };

  Scalar rng_scalar() {
    uint8_t b[64];
    rng_.rng_fill(b, 64);
    return Scalar::from_wide(b);
  }
  uint32_t G(size_t i) const { return (uint32_t)(2 + i); }
  uint32_t H(size_t i) const { return (uint32_t)(2 + cap_ + i); }
  static Scalar inner(const std::vector<Scalar>& a, size_t ao, const std::vector<Scalar>& b, size_t bo, size_t n) {
    Scalar acc = Scalar::zero();
    for (size_t i = 0; i < n; ++i) acc += a[ao + i] * b[bo + i];
    return acc;
  }
  // blind * B_blinding + <a, G[off..]> + <b, H[off..]>
  MsmRow vec_row(const Scalar& blind, const std::vector<Scalar>* a, const std::vector<Scalar>* b, size_t off, size_t n) const {
    MsmRow r;
    r.add(blind, 1);
    if (a) for (size_t i = 0; i < n; ++i) r.add((*a)[off + i], G(off + i));
    if (b) for (size_t i = 0; i < n; ++i) r.add((*b)[off + i], H(off + i));
    return r;
  }

  void after_commitments(const uint8_t* pts, std::vector<MsmRow>& rows) {
    const size_t m = cs_.v.size();
    commitments_.assign(pts, pts + 32 * m);
    for (size_t i = 0; i < m; ++i) cs_.tr.append_point("V", pts + 32 * i);
    // constraints + first-phase witness
    synth_(cs_, vars_);
    if (cs_.failed || !cs_.queue.empty()) { failed_ = true; stage_ = kDone; return; }
    cs_.tr.append_u64("m", m);
    rng_ = cs_.tr;
    for (size_t i = 0; i < m; ++i) {
      uint8_t b[32];
      cs_.v_bl[i].to_bytes(b);
      rng_.rekey_with_witness("v_blinding", b, 32);
    }
    uint8_t rng_seed[32];
    derive(seed_, "rng", 0, rng_seed, 32);
    rng_.finalize_rng(rng_seed);
    n1_ = cs_.n_vars();
    i_bl1_ = rng_scalar(); o_bl1_ = rng_scalar(); s_bl1_ = rng_scalar();
    sL_.resize(n1_); sR_.resize(n1_);
    for (size_t i = 0; i < n1_; ++i) sL_[i] = rng_scalar();
    for (size_t i = 0; i < n1_; ++i) sR_[i] = rng_scalar();
    if (n1_ > cap_) { failed_ = true; stage_ = kDone; return; }
    rows.push_back(vec_row(i_bl1_, &cs_.aL, &cs_.aR, 0, n1_));
    rows.push_back(vec_row(o_bl1_, &cs_.aO, nullptr, 0, n1_));
    rows.push_back(vec_row(s_bl1_, &sL_, &sR_, 0, n1_));
    stage_ = kPhase1;
  }

  void after_phase1(const uint8_t* pts, std::vector<MsmRow>& rows) {
    std::memcpy(head_, pts, 96);
    cs_.tr.append_point("A_I1", pts);
    cs_.tr.append_point("A_O1", pts + 32);
    cs_.tr.append_point("S1", pts + 64);
    if (!cs_.deferred()) {
      cs_.tr.append_message("dom-sep", (const uint8_t*)"r1cs-1phase", 11);
    } else {
      cs_.tr.append_message("dom-sep", (const uint8_t*)"r1cs-2phase", 11);
      cs_.second_phase();
    }
    n_ = cs_.n_vars();
    pn_ = 1; k_ = 0;
    while (pn_ < n_) { pn_ <<= 1; ++k_; }
    if (pn_ > cap_ || cs_.failed) { failed_ = true; stage_ = kDone; return; }
    const size_t n2 = n_ - n1_;
    i_bl2_ = o_bl2_ = s_bl2_ = Scalar::zero();
    sL_.resize(n_); sR_.resize(n_);
    if (n2 > 0) {
      i_bl2_ = rng_scalar(); o_bl2_ = rng_scalar(); s_bl2_ = rng_scalar();
      for (size_t i = n1_; i < n_; ++i) sL_[i] = rng_scalar();
      for (size_t i = n1_; i < n_; ++i) sR_[i] = rng_scalar();
      rows.push_back(vec_row(i_bl2_, &cs_.aL, &cs_.aR, n1_, n2));
      rows.push_back(vec_row(o_bl2_, &cs_.aO, nullptr, n1_, n2));
      rows.push_back(vec_row(s_bl2_, &sL_, &sR_, n1_, n2));
    } else {
      rows.resize(3);   // three empty rows: the identity, whose encoding is all zeros
    }
    stage_ = kPhase2;
  }

  void after_phase2(const uint8_t* pts, std::vector<MsmRow>& rows) {
    std::memcpy(head_ + 96, pts, 96);
    cs_.tr.append_point("A_I2", pts);
    cs_.tr.append_point("A_O2", pts + 32);
    cs_.tr.append_point("S2", pts + 64);
    y_ = cs_.tr.challenge_scalar("y");
    const Scalar z = cs_.tr.challenge_scalar("z");
    const size_t n = n_, m = cs_.v.size();
    std::vector<Scalar> wL(n, Scalar::zero()), wR(n, Scalar::zero()), wO(n, Scalar::zero());
    wV_.assign(m, Scalar::zero());
    Scalar exp_z = z;
    for (const auto& lc : cs_.constraints()) {
      for (const auto& term : lc.terms) {
        const Scalar t = exp_z * term.second;
        switch (term.first.kind) {
          case VarKind::MulLeft: wL[term.first.idx] += t; break;
          case VarKind::MulRight: wR[term.first.idx] += t; break;
          case VarKind::MulOut: wO[term.first.idx] += t; break;
          case VarKind::Committed: wV_[term.first.idx] -= t; break;
          case VarKind::One: break;
        }
      }
      exp_z *= z;
    }
    // l(x) = l1 x + l2 x^2 + l3 x^3,  r(x) = r0 + r1 x + r3 x^3
    l1_.assign(n, Scalar::zero()); l2_ = l1_; l3_ = l1_; r0_ = l1_; r1_ = l1_; r3_ = l1_;
    yinv_pow_.assign(pn_, Scalar::one());
    const Scalar y_inv = y_.invert();
    for (size_t i = 1; i < pn_; ++i) yinv_pow_[i] = yinv_pow_[i - 1] * y_inv;
    exp_y_ = Scalar::one();
    for (size_t i = 0; i < n; ++i) {
      l1_[i] = cs_.aL[i] + yinv_pow_[i] * wR[i];
      l2_[i] = cs_.aO[i];
      l3_[i] = sL_[i];
      r0_[i] = wO[i] - exp_y_;
      r1_[i] = exp_y_ * cs_.aR[i] + wL[i];
      r3_[i] = exp_y_ * sR_[i];
      exp_y_ *= y_;
    }
    t_[1] = inner(l1_, 0, r0_, 0, n);
    t_[2] = inner(l1_, 0, r1_, 0, n) + inner(l2_, 0, r0_, 0, n);
    t_[3] = inner(l2_, 0, r1_, 0, n) + inner(l3_, 0, r0_, 0, n);
    t_[4] = inner(l1_, 0, r3_, 0, n) + inner(l3_, 0, r1_, 0, n);
    t_[5] = inner(l2_, 0, r3_, 0, n);
    t_[6] = inner(l3_, 0, r3_, 0, n);
    tb_[2] = Scalar::zero();
    const int order[5] = {1, 3, 4, 5, 6};
    for (int i : order) tb_[i] = rng_scalar();
    for (int i : order) {
      MsmRow r;
      r.add(t_[i], 0);
      r.add(tb_[i], 1);
      rows.push_back(std::move(r));
    }
    stage_ = kT;
  }

  void after_t(const uint8_t* pts, std::vector<MsmRow>& rows) {
    std::memcpy(head_ + 192, pts, 160);
    const char* labels[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
    for (int i = 0; i < 5; ++i) cs_.tr.append_point(labels[i], pts + 32 * i);
    const Scalar u = cs_.tr.challenge_scalar("u");
    const Scalar x = cs_.tr.challenge_scalar("x");
    const size_t n = n_, m = cs_.v.size();
    for (size_t i = 0; i < m; ++i) tb_[2] += wV_[i] * cs_.v_bl[i];
    Scalar xp[7];
    xp[0] = Scalar::one();
    for (int i = 1; i <= 6; ++i) xp[i] = xp[i - 1] * x;
    Scalar t_x = Scalar::zero(), t_x_bl = Scalar::zero();
    for (int i = 1; i <= 6; ++i) { t_x += t_[i] * xp[i]; t_x_bl += tb_[i] * xp[i]; }
    lv_.assign(pn_, Scalar::zero());
    rv_.assign(pn_, Scalar::zero());
    for (size_t i = 0; i < n; ++i) {
      lv_[i] = l1_[i] * xp[1] + l2_[i] * xp[2] + l3_[i] * xp[3];
      rv_[i] = r0_[i] + r1_[i] * xp[1] + r3_[i] * xp[3];
    }
    for (size_t i = n; i < pn_; ++i) { rv_[i] = -exp_y_; exp_y_ *= y_; }
    const Scalar i_bl = i_bl1_ + u * i_bl2_, o_bl = o_bl1_ + u * o_bl2_, s_bl = s_bl1_ + u * s_bl2_;
    const Scalar e_bl = ((x * s_bl + o_bl) * x + i_bl) * x;
    cs_.tr.append_scalar("t_x", t_x);
    cs_.tr.append_scalar("t_x_blinding", t_x_bl);
    cs_.tr.append_scalar("e_blinding", e_bl);
    w_ = cs_.tr.challenge_scalar("w");
    cG_.resize(pn_); cH_.resize(pn_);
    for (size_t i = 0; i < pn_; ++i) {
      cG_[i] = i < n1_ ? Scalar::one() : u;
      cH_[i] = yinv_pow_[i] * cG_[i];
    }
    proof_.clear();
    proof_.push_back(1);   // two-phase wire format
    proof_.insert(proof_.end(), head_, head_ + 352);
    uint8_t b[32];
    t_x.to_bytes(b); proof_.insert(proof_.end(), b, b + 32);
    t_x_bl.to_bytes(b); proof_.insert(proof_.end(), b, b + 32);
    e_bl.to_bytes(b); proof_.insert(proof_.end(), b, b + 32);
    cs_.tr.append_message("dom-sep", (const uint8_t*)"ipp v1", 6);
    cs_.tr.append_u64("n", pn_);
    len_ = pn_;
    round_ = 0;
    if (k_ == 0) { finish(); return; }
    if (!device_ipa_) ipa_rows(rows);
    stage_ = kIpa;
  }

  // L and R of the current round as multiscalar multiplications over the ORIGINAL generators
  void ipa_rows(std::vector<MsmRow>& rows) {
    const size_t half = len_ / 2;
    const Scalar cL = inner(lv_, 0, rv_, half, half), cR = inner(lv_, half, rv_, 0, half);
    MsmRow L, R;
    for (size_t idx = 0; idx < pn_; ++idx) {
      const size_t j = idx % len_;
      const bool hi = j >= half;
      const size_t jj = hi ? j - half : j;
      if (hi) { L.add(lv_[jj] * cG_[idx], G(idx)); R.add(rv_[jj] * cH_[idx], H(idx)); }        // a_L G_R ; b_L H_R
      else { L.add(rv_[half + jj] * cH_[idx], H(idx)); R.add(lv_[half + jj] * cG_[idx], G(idx)); }  // b_R H_L ; a_R G_L
    }
    L.add(cL * w_, 0);   // Q = w B
    R.add(cR * w_, 0);
    rows.push_back(std::move(L));
    rows.push_back(std::move(R));
  }

  void after_ipa_round(const uint8_t* pts, std::vector<MsmRow>& rows) {
    proof_.insert(proof_.end(), pts, pts + 64);
    cs_.tr.append_point("L", pts);
    cs_.tr.append_point("R", pts + 32);
    const Scalar uu = cs_.tr.challenge_scalar("u"), uu_inv = uu.invert();
    const size_t half = len_ / 2;
    for (size_t j = 0; j < half; ++j) {
      lv_[j] = lv_[j] * uu + lv_[half + j] * uu_inv;
      rv_[j] = rv_[j] * uu_inv + rv_[half + j] * uu;
    }
    for (size_t idx = 0; idx < pn_; ++idx) {
      const bool hi = (idx % len_) >= half;
      cG_[idx] *= hi ? uu : uu_inv;
      cH_[idx] *= hi ? uu_inv : uu;
    }
    len_ = half;
    if (++round_ == k_) { finish(); return; }
    ipa_rows(rows);
  }

  void finish() {
    uint8_t b[32];
    lv_[0].to_bytes(b); proof_.insert(proof_.end(), b, b + 32);
    rv_[0].to_bytes(b); proof_.insert(proof_.end(), b, b + 32);
    stage_ = kDone;
  }

  R1csProverCS cs_;
  Transcript rng_{"unused"};
  size_t cap_;
  uint8_t seed_[32];
  std::vector<Scalar> values_, blindings_;
  Synthesis synth_;
  std::vector<Var> vars_;
  std::vector<uint8_t> commitments_, proof_;
  uint8_t head_[352];   // A_I1 A_O1 S1 A_I2 A_O2 S2 T_1 T_3 T_4 T_5 T_6
  Stage stage_ = kNew;
  bool failed_ = false, device_ipa_ = false;
  size_t n1_ = 0, n_ = 0, pn_ = 1, k_ = 0, len_ = 0, round_ = 0;
  Scalar i_bl1_, o_bl1_, s_bl1_, i_bl2_, o_bl2_, s_bl2_, y_, exp_y_, w_;
  Scalar t_[7], tb_[7];
  std::vector<Scalar> sL_, sR_, wV_, l1_, l2_, l3_, r0_, r1_, r3_, yinv_pow_, lv_, rv_, cG_, cH_;
};

// The ZkVM cloak: n_in + n_out values (quantity, flavor), commitment blindings derived from the seed as the
// oracle's zko_cloak_prove does ("q_blinding" / "f_blinding", value index).
inline std::unique_ptr<R1csProver> cloak_prover(size_t n_in, size_t n_out, const uint64_t* quantities, const uint8_t* flavors,
                                                const uint8_t seed[32], size_t gens_capacity) {
  std::vector<cloak::Amount> amounts;
  std::vector<Scalar> values, blindings;
  for (size_t i = 0; i < n_in + n_out; ++i) {
    cloak::Amount a;
    a.q = quantities[i];
    uint8_t wide[64] = {0};
    std::memcpy(wide, flavors + 32 * i, 32);
    a.f = Scalar::from_wide(wide);
    amounts.push_back(a);
    values.push_back(Scalar::from_u64(a.q)); blindings.push_back(R1csProver::derive_scalar(seed, "q_blinding", i));
    values.push_back(a.f); blindings.push_back(R1csProver::derive_scalar(seed, "f_blinding", i));
  }
  auto synth = [amounts, n_in](R1csProverCS& cs, const std::vector<Var>& vars) {
    std::vector<cloak::Amount> in(amounts.begin(), amounts.begin() + n_in), out(amounts.begin() + n_in, amounts.end());
    cloak::gadget_witness(in, out, cs.queue);
    std::vector<Value> vals;
    for (size_t i = 0; i + 1 < vars.size(); i += 2) vals.push_back(Value{vars[i], vars[i + 1]});
    std::vector<Value> vin(vals.begin(), vals.begin() + n_in), vout(vals.begin() + n_in, vals.end());
    cloak::gadget(cs, vin, vout);
  };
  return std::unique_ptr<R1csProver>(new R1csProver("ZkVM.r1cs", seed, gens_capacity, std::move(values), std::move(blindings), synth));
}

// A constraint system described as data (R1csDesc): the prover needs, beyond the verifier's description, the
// WITNESS -- the committed values, and for every multiplier either its (left, right) assignment (`given`, in index
// order, for multipliers allocated with explicit values) or the two constraints that define it (mult_def[2 i],
// mult_def[2 i + 1]: the constraints `left - l_i = 0` / `right - r_i = 0` a multiply() emitted; 0xffffffff = given).
// Defined multipliers are evaluated in index order, second-phase ones once the challenges are drawn.
constexpr uint32_t MULT_GIVEN = 0xffffffffu;

inline std::unique_ptr<R1csProver> desc_prover(const R1csDesc& d, const std::vector<uint32_t>& mult_def, std::vector<Scalar> values,
                                               std::vector<Scalar> blindings, const std::vector<std::pair<Scalar, Scalar>>& given,
                                               const uint8_t seed[32], size_t gens_capacity) {
  auto synth = [d, mult_def, given](R1csProverCS& cs, const std::vector<Var>&) {
    auto next_given = std::make_shared<size_t>(0);
    R1csProverCS* pcs = &cs;
    auto coef_of = [](const R1csDesc::Term& t, const std::vector<Scalar>& chal) {
      Scalar c = t.c;
      if (t.chal >= 0) for (uint32_t e = 0; e < t.pow; ++e) c *= chal[(size_t)t.chal];
      return c;
    };
    // value of the multiplier side (kind, i) from its defining constraint  sum_others + coef * side = 0
    auto solve = [pcs, coef_of, &d](uint32_t con_idx, VarKind kind, uint32_t i, const std::vector<Scalar>& chal, bool& ok) {
      Scalar acc = Scalar::zero(), own = Scalar::zero();
      if (con_idx >= d.cons.size()) { ok = false; return acc; }
      for (const auto& t : d.cons[con_idx]) {
        const Scalar c = coef_of(t, chal);
        if (t.kind == kind && t.idx == i) { own += c; continue; }
        Scalar val = Scalar::one();
        const size_t have = pcs->aL.size();
        switch (t.kind) {
          case VarKind::MulLeft: if (t.idx >= have) { ok = false; return acc; } val = pcs->aL[t.idx]; break;
          case VarKind::MulRight: if (t.idx >= have) { ok = false; return acc; } val = pcs->aR[t.idx]; break;
          case VarKind::MulOut: if (t.idx >= have) { ok = false; return acc; } val = pcs->aO[t.idx]; break;
          case VarKind::Committed: if (t.idx >= pcs->v.size()) { ok = false; return acc; } val = pcs->v[t.idx]; break;
          case VarKind::One: break;
        }
        acc += c * val;
      }
      if (own == Scalar::zero()) { ok = false; return acc; }
      return own == -Scalar::one() ? acc : acc * (-own).invert();
    };
    auto assign = [pcs, solve, next_given, &mult_def, &given](uint32_t i, const std::vector<Scalar>& chal) {
      Scalar l, r;
      bool ok = true;
      const uint32_t dl = 2 * i < mult_def.size() ? mult_def[2 * i] : MULT_GIVEN, dr = 2 * i + 1 < mult_def.size() ? mult_def[2 * i + 1] : MULT_GIVEN;
      if (dl == MULT_GIVEN || dr == MULT_GIVEN) {
        if (*next_given >= given.size()) { pcs->failed = true; l = r = Scalar::zero(); }
        else { l = given[*next_given].first; r = given[*next_given].second; ++*next_given; }
      } else {
        l = solve(dl, VarKind::MulLeft, i, chal, ok);
        r = solve(dr, VarKind::MulRight, i, chal, ok);
        if (!ok) pcs->failed = true;
      }
      pcs->queue.emplace_back(l, r);
      Var o[3];
      pcs->allocate_multiplier(o);
    };
    auto add_all = [coef_of, &d](ConstraintSystemT<Scalar>& c, const std::vector<Scalar>& chal) {
      for (const auto& con : d.cons) {
        LCt<Scalar> lc;
        for (const auto& t : con) lc.add(Var{t.kind, t.idx}, coef_of(t, chal));
        c.constrain(std::move(lc));
      }
    };
    for (uint32_t i = 0; i < d.n1; ++i) assign(i, {});
    if (d.chal_names.empty() && d.n == d.n1) {
      add_all(cs, {});
    } else {
      cs.specify_randomized_constraints([assign, add_all, &d](ConstraintSystemT<Scalar>& c) {
        std::vector<Scalar> chal;
        for (const std::string& name : d.chal_names) chal.push_back(c.challenge_scalar(name.c_str()));
        for (uint32_t i = d.n1; i < d.n; ++i) assign(i, chal);
        add_all(c, chal);
      });
    }
  };
  return std::unique_ptr<R1csProver>(new R1csProver(d.label.c_str(), seed, gens_capacity, std::move(values), std::move(blindings), synth));
}

}  // namespace zk
