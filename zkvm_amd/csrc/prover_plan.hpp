// prover_plan.hpp -- host side of the device prover (prover_dev.hpp): a described constraint system + the
// definitions of its multipliers -> the constant tables the phase kernels read, the layout of the per-proof state
// and the scaffolding (offsets, generator indices) of the multiscalar multiplications between the phases.
#pragma once
#include "cloak_plan.hpp"
#include "prover_dev.hpp"
#include "r1cs_prover.hpp"

#include <string>
#include <vector>

namespace zk {

struct PvHostPlan {
  PvShape sh{};
  std::vector<uint32_t> init, mono_chal, mono_pow, con_off, t_kind, t_idx, t_mono, t_coef, mult_def, given_slot, tgt_off,
      term_info, prod_qm, prod_coef;
  std::vector<uint8_t> chal_labels;

  PvPlan view() const {   // host pointers (the emulation); the library uploads the vectors and fills a PvPlan of device pointers
    PvPlan p;
    p.init = init.data(); p.chal_labels = chal_labels.data(); p.mono_chal = mono_chal.data(); p.mono_pow = mono_pow.data();
    p.con_off = con_off.data(); p.t_kind = t_kind.data(); p.t_idx = t_idx.data(); p.t_mono = t_mono.data(); p.t_coef = t_coef.data();
    p.mult_def = mult_def.data(); p.given_slot = given_slot.data(); p.tgt_off = tgt_off.data(); p.term_info = term_info.data();
    p.prod_qm = prod_qm.data(); p.prod_coef = prod_coef.data();
    return p;
  }
};

// throws std::runtime_error for a description the device prover cannot serve
inline PvHostPlan pv_build(const R1csDesc& d, const std::vector<uint32_t>& mult_def, size_t gens_capacity) {
  PvHostPlan hp;
  const CloakPlan p = plan_from_desc(d);
  PvShape& sh = hp.sh;
  sh.m = p.m; sh.n1 = p.n1; sh.n = p.n; sh.pn = p.pn; sh.k = p.k; sh.n_cons = p.n_cons;
  sh.n_chal2 = (uint32_t)d.chal_names.size();
  sh.n_mono = (uint32_t)p.mono_chal.size();
  sh.gens_capacity = (uint32_t)gens_capacity;
  sh.two_phase = (d.chal_names.empty() && d.n == d.n1) ? 0u : 1u;
  if (p.pn > gens_capacity) throw std::runtime_error("prover: statement needs more generators than the set holds");
  if (mult_def.size() != 2 * (size_t)d.n) throw std::runtime_error("prover: mult_def must hold two entries per multiplier");
  hp.mono_chal = p.mono_chal; hp.mono_pow = p.mono_pow;
  hp.tgt_off = p.tgt_off; hp.term_info = p.term_info; hp.prod_coef = p.prod_coef;
  for (size_t i = 0; i < p.prod_q.size(); ++i) { hp.prod_qm.push_back(p.prod_q[i]); hp.prod_qm.push_back(p.prod_mono[i]); }
  auto mono_id = [&](int chal, uint32_t pow) -> uint32_t {
    if (chal < 0 || pow == 0) return 0;
    for (size_t i = 1; i < p.mono_chal.size(); ++i)
      if (p.mono_chal[i] == (uint32_t)chal && p.mono_pow[i] == pow) return (uint32_t)i;
    throw std::runtime_error("prover: monomial missing from the plan");
  };
  hp.con_off.push_back(0);
  for (const auto& con : d.cons) {
    for (const auto& t : con) {
      hp.t_kind.push_back((uint32_t)t.kind);
      hp.t_idx.push_back(t.idx);
      hp.t_mono.push_back(mono_id(t.chal, t.pow));
      uint8_t bytes[32];
      t.c.to_bytes(bytes);
      uint32_t w[8];
      for (int i = 0; i < 8; ++i) w[i] = (uint32_t)bytes[4 * i] | ((uint32_t)bytes[4 * i + 1] << 8) | ((uint32_t)bytes[4 * i + 2] << 16) | ((uint32_t)bytes[4 * i + 3] << 24);
      const scm mc = scm_from_words(w);
      for (int i = 0; i < 8; ++i) hp.t_coef.push_back(mc.v[i]);
    }
    hp.con_off.push_back((uint32_t)hp.t_kind.size());
  }
  hp.mult_def = mult_def;
  uint32_t given = 0;
  for (uint32_t i = 0; i < d.n; ++i) {
    const bool is_given = mult_def[2 * i] == PV_GIVEN || mult_def[2 * i + 1] == PV_GIVEN;
    hp.given_slot.push_back(is_given ? given++ : PV_GIVEN);
    if (!is_given && (mult_def[2 * i] >= p.n_cons || mult_def[2 * i + 1] >= p.n_cons)) throw std::runtime_error("prover: defining constraint out of range");
  }
  sh.n_given = given;
  for (const std::string& name : d.chal_names) {
    if (name.size() > PV_MAX_LABEL) throw std::runtime_error("prover: challenge label longer than 31 bytes");
    uint8_t slot[32] = {0};
    slot[0] = (uint8_t)name.size();
    std::memcpy(slot + 1, name.data(), name.size());
    hp.chal_labels.insert(hp.chal_labels.end(), slot, slot + 32);
  }
  if (hp.chal_labels.empty()) hp.chal_labels.assign(32, 0);
  {
    Transcript tr(d.label.c_str());
    tr.append_message("dom-sep", (const uint8_t*)"r1cs v1", 7);
    hp.init.resize(52);
    tr.export_state(hp.init.data());
  }
  sh.proof_len = 1 + 32 * (16 + 2 * sh.k);
  sh.proof_stride = (sh.proof_len + 15u) & ~15u;
  const uint32_t n2 = sh.n - sh.n1;
  sh.r1_terms = 3 + 5 * sh.n1;
  sh.r2_terms = n2 ? 3 + 5 * n2 : 0;
  // state layout (words)
  uint32_t o = 0;
  auto take = [&o](uint32_t words) { const uint32_t at = o; o += (words + 3u) & ~3u; return at; };
  sh.o_tr = take(52); sh.o_rng = take(52);
  sh.o_v = take(8 * sh.m); sh.o_vbl = take(8 * sh.m);
  sh.o_aL = take(8 * sh.n); sh.o_aR = take(8 * sh.n); sh.o_aO = take(8 * sh.n); sh.o_sL = take(8 * sh.n); sh.o_sR = take(8 * sh.n);
  sh.o_blind = take(8 * PV_BLIND_SLOTS); sh.o_chal = take(8 * PV_CHAL_SLOTS);
  sh.o_c2 = take(8 * std::max<uint32_t>(sh.n_chal2, 1)); sh.o_sym = take(8 * sh.n_mono);
  sh.o_wL = take(8 * sh.n); sh.o_wR = take(8 * sh.n); sh.o_wO = take(8 * sh.n); sh.o_wV = take(8 * std::max<uint32_t>(sh.m, 1));
  sh.o_t = take(8 * 7); sh.o_tb = take(8 * 7);
  sh.o_zpow = take(8 * (sh.n_cons + 1)); sh.o_ypow = take(8 * sh.pn); sh.o_yinv = take(8 * sh.pn);
  sh.o_flag = take(4);
  sh.state_words = o;
  return hp;
}

// rows of the multiscalar multiplications between the phases, for `batch` proofs: offsets (rows + 1) and generator
// indices (one per scalar); phase 0: batch x m rows (v B + blinding B_blinding), phases 1 / 2: A_I, A_O, S over the
// multipliers [first, last) (phase 2 without second-phase multipliers: three EMPTY rows, as the reference's prover),
// phase 3: five rows (t_i B + blinding B_blinding)
struct PvRows {
  std::vector<uint64_t> offsets;
  std::vector<uint32_t> index;
};
inline PvRows pv_rows_pairs(size_t rows) {
  PvRows r;
  for (size_t i = 0; i <= rows; ++i) r.offsets.push_back(2 * i);
  for (size_t i = 0; i < rows; ++i) { r.index.push_back(0); r.index.push_back(1); }
  return r;
}
inline PvRows pv_rows_commit(size_t batch, uint32_t first, uint32_t last, size_t cap, bool empty_when_none) {
  PvRows r;
  const uint32_t cnt = last - first;
  r.offsets.push_back(0);
  for (size_t b = 0; b < batch; ++b) {
    if (cnt == 0 && empty_when_none) { for (int k = 0; k < 3; ++k) r.offsets.push_back(r.offsets.back()); continue; }   // the identity
    r.index.push_back(1);
    for (uint32_t j = 0; j < cnt; ++j) r.index.push_back(2 + first + j);
    for (uint32_t j = 0; j < cnt; ++j) r.index.push_back((uint32_t)(2 + cap + first + j));
    r.offsets.push_back(r.offsets.back() + 1 + 2 * cnt);
    r.index.push_back(1);
    for (uint32_t j = 0; j < cnt; ++j) r.index.push_back(2 + first + j);
    r.offsets.push_back(r.offsets.back() + 1 + cnt);
    r.index.push_back(1);
    for (uint32_t j = 0; j < cnt; ++j) r.index.push_back(2 + first + j);
    for (uint32_t j = 0; j < cnt; ++j) r.index.push_back((uint32_t)(2 + cap + first + j));
    r.offsets.push_back(r.offsets.back() + 1 + 2 * cnt);
  }
  return r;
}

// The ZkVM cloak as a described system with the definitions of its multipliers: the gadget traced once with symbolic
// scalars (PlanBuilder), every multiply() noting the two constraints it emitted.
class PvCloakTrace : public ConstraintSystemT<SymScalar> {
 public:
  SymScalar challenge_scalar(const char* label) override {
    labels_.push_back(label);
    return SymScalar{Scalar::one(), (int)labels_.size() - 1, 1};
  }
  void multiply(LC left, LC right, Var out[3]) override {
    const uint32_t i = num_vars_, q = (uint32_t)cons_.size();
    ConstraintSystemT<SymScalar>::multiply(std::move(left), std::move(right), out);
    if (def_.size() < 2 * (size_t)(i + 1)) def_.resize(2 * (size_t)(i + 1), PV_GIVEN);
    def_[2 * i] = q;
    def_[2 * i + 1] = q + 1;
  }
  static void trace(uint32_t n_in, uint32_t n_out, R1csDesc& d, std::vector<uint32_t>& mult_def) {
    PvCloakTrace b;
    d = R1csDesc();
    d.m = 2 * (n_in + n_out);
    const std::vector<Value> vals = cloak::committed_values(n_in + n_out);
    std::vector<Value> in(vals.begin(), vals.begin() + n_in), out(vals.begin() + n_in, vals.end());
    cloak::gadget(b, in, out);
    d.n1 = (uint32_t)b.run_second_phase();
    d.n = (uint32_t)b.num_vars_;
    d.chal_names = b.labels_;
    for (const auto& lc : b.cons_) {
      std::vector<R1csDesc::Term> con;
      for (const auto& term : lc.terms) con.push_back(R1csDesc::Term{term.first.kind, term.first.idx, term.second.c, term.second.chal, (uint32_t)term.second.pow});
      d.cons.push_back(std::move(con));
    }
    mult_def = b.def_;
    mult_def.resize(2 * (size_t)d.n, PV_GIVEN);
  }

 private:
  std::vector<std::string> labels_;
  std::vector<uint32_t> def_;
};

// the (left, right) assignments the cloak hands in: what cloak::gadget_witness queues, as canonical words
inline void pv_cloak_given(uint32_t n_in, uint32_t n_out, const uint64_t* quantities, const uint8_t* flavors, std::vector<uint32_t>& out) {
  std::vector<cloak::Amount> amounts;
  for (size_t i = 0; i < (size_t)n_in + n_out; ++i) {
    cloak::Amount a;
    a.q = quantities[i];
    uint8_t wide[64] = {0};
    std::memcpy(wide, flavors + 32 * i, 32);
    a.f = Scalar::from_wide(wide);
    amounts.push_back(a);
  }
  std::vector<cloak::Amount> in(amounts.begin(), amounts.begin() + n_in), outv(amounts.begin() + n_in, amounts.end());
  std::deque<std::pair<Scalar, Scalar>> q;
  cloak::gadget_witness(in, outv, q);
  for (const auto& lr : q) {
    uint8_t b[64];
    lr.first.to_bytes(b);
    lr.second.to_bytes(b + 32);
    for (int i = 0; i < 16; ++i) out.push_back((uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24));
  }
}

}  // namespace zk
