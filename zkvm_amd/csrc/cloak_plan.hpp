// cloak_plan.hpp -- compile the constraint system of a cloak statement into a
// "plan" the device replays per transaction (host side).
//
// Every transaction of one shape (n_in inputs, n_out outputs) has the same
// constraint system; only the challenge values differ.  Running the gadget once
// with symbolic scalars (coefficient = constant * challenge^power) yields, per
// flattening target (wL_i, wR_i, wO_i, wV_j, wc), the list of terms
//     (constraint index q, coefficient constant, monomial)
// and the device computes  target = sum const * monomial(challenges) * z^(q+1).
#pragma once
#include "r1cs_verifier.hpp"
#include "sc_dev.hpp"

#include <stdexcept>
#include <string>
#include <vector>

namespace zk {

struct SymScalar {   // c * challenge[chal]^pow  (chal < 0: a constant)
  Scalar c;
  int chal = -1, pow = 0;
  static SymScalar one() { return SymScalar{Scalar::one(), -1, 0}; }
  static SymScalar zero() { return SymScalar{Scalar::zero(), -1, 0}; }
  SymScalar operator-() const { return SymScalar{-c, chal, pow}; }
  friend SymScalar operator*(const SymScalar& a, const SymScalar& b) {
    if (a.chal >= 0 && b.chal >= 0 && a.chal != b.chal)
      throw std::runtime_error("cloak plan: product of two different challenges is not supported");
    return SymScalar{a.c * b.c, a.chal >= 0 ? a.chal : b.chal, a.pow + b.pow};
  }
  friend SymScalar operator+(const SymScalar& a, const SymScalar& b) {
    if (a.chal != b.chal || a.pow != b.pow) throw std::runtime_error("cloak plan: sum of different monomials");
    return SymScalar{a.c + b.c, a.chal, a.pow};
  }
};

struct CloakPlan {
  uint32_t n_in = 0, n_out = 0, m = 0, n1 = 0, n = 0, pn = 0, k = 0, n_cons = 0;
  std::vector<uint8_t> chal_label;            // per second-phase challenge: 0 mix, 1 k-value shuffle, 2 shuffle
  std::vector<uint32_t> mono_chal, mono_pow;  // monomial 0 is the constant 1 (chal = 0xffffffff)
  std::vector<uint32_t> tgt_off;              // 3n + m + 1 targets (+1): wL | wR | wO | wV | wc
  std::vector<uint32_t> term_q, term_mono;
  std::vector<uint32_t> term_coef;            // 8 words per term, Montgomery form, sign folded in
  uint32_t n_targets() const { return 3 * n + m + 1; }
};

class PlanBuilder : public ConstraintSystemT<SymScalar> {
 public:
  SymScalar challenge_scalar(const char* label) override {
    const std::string s(label);
    uint8_t id;
    if (s == "mix challenge") id = 0;
    else if (s == "k-value shuffle challenge") id = 1;
    else if (s == "shuffle challenge") id = 2;
    else throw std::runtime_error("cloak plan: unknown challenge label " + s);
    labels_.push_back(id);
    return SymScalar{Scalar::one(), (int)labels_.size() - 1, 1};
  }

  static CloakPlan build(uint32_t n_in, uint32_t n_out) {
    PlanBuilder b;
    CloakPlan p;
    p.n_in = n_in; p.n_out = n_out; p.m = 2 * (n_in + n_out);
    const std::vector<Value> vals = cloak::committed_values(n_in + n_out);
    std::vector<Value> in(vals.begin(), vals.begin() + n_in), out(vals.begin() + n_in, vals.end());
    cloak::gadget(b, in, out);
    p.n1 = (uint32_t)b.run_second_phase();
    p.n = (uint32_t)b.num_vars_;
    p.pn = 1; p.k = 0;
    while (p.pn < p.n) { p.pn <<= 1; ++p.k; }
    p.n_cons = (uint32_t)b.cons_.size();
    p.chal_label = b.labels_;
    // monomials
    p.mono_chal.push_back(0xffffffffu); p.mono_pow.push_back(0);
    auto mono_id = [&](int chal, int pow) -> uint32_t {
      if (chal < 0 || pow == 0) return 0;
      for (size_t i = 1; i < p.mono_chal.size(); ++i)
        if (p.mono_chal[i] == (uint32_t)chal && p.mono_pow[i] == (uint32_t)pow) return (uint32_t)i;
      p.mono_chal.push_back((uint32_t)chal); p.mono_pow.push_back((uint32_t)pow);
      return (uint32_t)p.mono_chal.size() - 1;
    };
    // bucket terms by target
    const uint32_t T = p.n_targets();
    struct Term { uint32_t q, mono; Scalar c; };
    std::vector<std::vector<Term>> by_tgt(T);
    for (uint32_t q = 0; q < p.n_cons; ++q) {
      for (const auto& term : b.cons_[q].terms) {
        uint32_t tgt; bool negate = false;
        switch (term.first.kind) {
          case VarKind::MulLeft: tgt = term.first.idx; break;
          case VarKind::MulRight: tgt = p.n + term.first.idx; break;
          case VarKind::MulOut: tgt = 2 * p.n + term.first.idx; break;
          case VarKind::Committed: tgt = 3 * p.n + term.first.idx; negate = true; break;
          default: tgt = 3 * p.n + p.m; negate = true; break;
        }
        const SymScalar& s = term.second;
        by_tgt[tgt].push_back(Term{q, mono_id(s.chal, s.pow), negate ? -s.c : s.c});
      }
    }
    p.tgt_off.assign(T + 1, 0);
    for (uint32_t t = 0; t < T; ++t) {
      p.tgt_off[t + 1] = p.tgt_off[t] + (uint32_t)by_tgt[t].size();
      for (const Term& tm : by_tgt[t]) {
        p.term_q.push_back(tm.q);
        p.term_mono.push_back(tm.mono);
        uint8_t bytes[32];
        tm.c.to_bytes(bytes);
        uint32_t w[8];
        for (int i = 0; i < 8; ++i) w[i] = (uint32_t)bytes[4 * i] | ((uint32_t)bytes[4 * i + 1] << 8) | ((uint32_t)bytes[4 * i + 2] << 16) | ((uint32_t)bytes[4 * i + 3] << 24);
        const scm mc = scm_from_words(w);
        for (int i = 0; i < 8; ++i) p.term_coef.push_back(mc.v[i]);
      }
    }
    return p;
  }

 private:
  std::vector<uint8_t> labels_;
};

}  // namespace zk
