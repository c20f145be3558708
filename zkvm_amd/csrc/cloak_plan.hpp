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

// A constraint system as data: what include/zkgpu.h calls zkgpu_r1cs_desc (the generic entry point of SURVEY.md
// sec 8 row f-3).  Constraint q is  sum_terms coefficient * var = 0  with coefficient = c * challenge[chal]^pow
// (chal < 0: the constant c); the order of `cons` is the order the constraint system emitted them (constraint q
// is weighted z^(q+1) in the flattening); multipliers [0, n1) belong to the first phase, [n1, n) to the second;
// chal_names are the Merlin labels of the second-phase challenges in the order they are drawn.
struct R1csDesc {
  std::string label = "ZkVM.r1cs";
  uint32_t m = 0, n1 = 0, n = 0;
  std::vector<std::string> chal_names;
  struct Term { VarKind kind; uint32_t idx; Scalar c; int chal; uint32_t pow; };
  std::vector<std::vector<Term>> cons;
};

struct CloakPlan {
  uint32_t n_in = 0, n_out = 0, m = 0, n1 = 0, n = 0, pn = 0, k = 0, n_cons = 0;
  std::string label;                          // transcript label
  std::vector<std::string> chal_names;        // second-phase challenge labels, in drawing order
  std::vector<uint32_t> mono_chal, mono_pow;  // monomial 0 is the constant 1 (chal = 0xffffffff)
  std::vector<uint32_t> tgt_off;              // 3n + m + 1 targets (+1): wL | wR | wO | wV | wc
  std::vector<uint32_t> term_q, term_mono;
  std::vector<uint32_t> term_coef;            // 8 words per term, Montgomery form, sign folded in
  // the same terms as k_prepare replays them: a UNIT term (coefficient +-1, no challenge) is read straight from the z
  // power table -- term_info = 0x80000000 | sign << 30 | q --, every other term is a product computed in a pass of its
  // own -- term_info = index into the product list (q, monomial, coefficient as ten 26-bit limbs, Montgomery form)
  std::vector<uint32_t> term_info, prod_q, prod_mono, prod_coef;
  std::vector<uint32_t> prod_off;             // per target: products before it (n_targets + 1)
  uint32_t n_targets() const { return 3 * n + m + 1; }
};

// description -> plan: monomial table, terms bucketed by flattening target (wL_i, wR_i, wO_i, wV_j, wc)
inline CloakPlan plan_from_desc(const R1csDesc& d) {
  CloakPlan p;
  p.label = d.label;
  p.m = d.m; p.n1 = d.n1; p.n = d.n;
  if (d.n1 > d.n) throw std::runtime_error("r1cs plan: more first-phase multipliers than multipliers");
  p.pn = 1; p.k = 0;
  while (p.pn < p.n) { p.pn <<= 1; ++p.k; }
  p.n_cons = (uint32_t)d.cons.size();
  p.chal_names = d.chal_names;
  p.mono_chal.push_back(0xffffffffu); p.mono_pow.push_back(0);
  auto mono_id = [&](int chal, uint32_t pow) -> uint32_t {
    if (chal < 0 || pow == 0) return 0;
    for (size_t i = 1; i < p.mono_chal.size(); ++i)
      if (p.mono_chal[i] == (uint32_t)chal && p.mono_pow[i] == pow) return (uint32_t)i;
    p.mono_chal.push_back((uint32_t)chal); p.mono_pow.push_back(pow);
    return (uint32_t)p.mono_chal.size() - 1;
  };
  const uint32_t T = p.n_targets();
  struct Term { uint32_t q, mono; Scalar c; };
  std::vector<std::vector<Term>> by_tgt(T);
  for (uint32_t q = 0; q < p.n_cons; ++q) {
    for (const auto& term : d.cons[q]) {
      uint32_t tgt; bool negate = false;
      if (term.chal >= (int)d.chal_names.size()) throw std::runtime_error("r1cs plan: challenge index out of range");
      switch (term.kind) {
        case VarKind::MulLeft: case VarKind::MulRight: case VarKind::MulOut:
          if (term.idx >= p.n) throw std::runtime_error("r1cs plan: multiplier index out of range");
          tgt = (term.kind == VarKind::MulLeft ? 0 : term.kind == VarKind::MulRight ? p.n : 2 * p.n) + term.idx;
          break;
        case VarKind::Committed:
          if (term.idx >= p.m) throw std::runtime_error("r1cs plan: commitment index out of range");
          tgt = 3 * p.n + term.idx; negate = true;
          break;
        default: tgt = 3 * p.n + p.m; negate = true; break;
      }
      by_tgt[tgt].push_back(Term{q, mono_id(term.chal, term.pow), negate ? -term.c : term.c});
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
      const bool plus = tm.c == Scalar::one(), minus = tm.c == -Scalar::one();
      if (tm.mono == 0 && (plus || minus)) {
        p.term_info.push_back(0x80000000u | (minus ? 0x40000000u : 0u) | tm.q);
      } else {
        p.term_info.push_back((uint32_t)p.prod_q.size());
        p.prod_q.push_back(tm.q);
        p.prod_mono.push_back(tm.mono);
        const scl lc = scl_from_scm(mc);
        for (int i = 0; i < 10; ++i) p.prod_coef.push_back(lc.v[i]);
      }
    }
    if (t == 0) p.prod_off.push_back(0);
    p.prod_off.push_back((uint32_t)p.prod_q.size());
  }
  if (p.n_cons >= (1u << 24) || p.prod_q.size() >= (1u << 24)) throw std::runtime_error("r1cs plan: too many constraints");
  return p;
}

// The numeric verifier of a described constraint system on the host (same R1csVerifier::prepare as the cloak):
// commitments = m x 32 bytes.  false: malformed proof.
inline bool prepare_desc(const R1csDesc& d, const uint8_t* commitments, const uint8_t* proof, size_t proof_len,
                         const Scalar& r, size_t gens_capacity, VerifierMsm& out) {
  R1csVerifier cs(d.label.c_str());
  for (uint32_t i = 0; i < d.m; ++i) cs.commit(commitments + 32 * i);
  Var o[3];
  for (uint32_t i = 0; i < d.n1; ++i) cs.allocate_multiplier(o);
  auto add_all = [&d](ConstraintSystemT<Scalar>& c, const std::vector<Scalar>& chal) {
    for (const auto& con : d.cons) {
      LCt<Scalar> lc;
      for (const auto& t : con) {
        Scalar coef = t.c;
        if (t.chal >= 0) for (uint32_t e = 0; e < t.pow; ++e) coef *= chal[(size_t)t.chal];
        lc.add(Var{t.kind, t.idx}, coef);
      }
      c.constrain(std::move(lc));
    }
  };
  if (d.chal_names.empty() && d.n == d.n1) {
    add_all(cs, {});
  } else {
    cs.specify_randomized_constraints([&d, add_all](ConstraintSystemT<Scalar>& c) {
      std::vector<Scalar> chal;
      for (const std::string& name : d.chal_names) chal.push_back(c.challenge_scalar(name.c_str()));
      Var oo[3];
      for (uint32_t i = d.n1; i < d.n; ++i) c.allocate_multiplier(oo);
      add_all(c, chal);
    });
  }
  return cs.prepare(proof, proof_len, r, gens_capacity, out);
}

class PlanBuilder : public ConstraintSystemT<SymScalar> {
 public:
  SymScalar challenge_scalar(const char* label) override {
    labels_.push_back(label);
    return SymScalar{Scalar::one(), (int)labels_.size() - 1, 1};
  }

  // the cloak gadget traced with symbolic scalars -> its description
  static R1csDesc describe(uint32_t n_in, uint32_t n_out) {
    PlanBuilder b;
    R1csDesc d;
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
    return d;
  }

  static CloakPlan build(uint32_t n_in, uint32_t n_out) {
    CloakPlan p = plan_from_desc(describe(n_in, n_out));
    p.n_in = n_in; p.n_out = n_out;
    return p;
  }

 private:
  std::vector<std::string> labels_;
};

}  // namespace zk
