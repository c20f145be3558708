// r1cs_verifier.hpp -- host side of bulletproofs::r1cs::Verifier: constraint
// collection, transcript replay, constraint flattening, inner-product
// verification scalars, and assembly of the one multiscalar multiplication the
// GPU evaluates (upstream
// `r1cs::Verifier::verify`, `InnerProductProof::verification_scalars`, and the
// `spacesuit::cloak` gadget that produces the constraints of a ZkVM `cloak`
// instruction -- none of whose Rust sources are mounted; written from Bunz et
// al. 2018 sec 3/5 and the crates' published protocol notes).
//
// The constraint system is generic in its scalar type S: with S = Scalar it is
// the numeric verifier (R1csVerifier); with S = SymScalar (cloak_plan.hpp) the
// same gadget code records a symbolic "tape" that the device-side preparation
// kernels replay per transaction.
//
// The output layout is the argument list of dalek's `mega_check`:
//   dynamic terms  [A_I1 A_O1 S1 A_I2 A_O2 S2 | V_0..V_{m-1} | T_1 T_3 T_4 T_5 T_6 | L_0.. | R_0..]
//   static terms   [B, B_blinding, G_0..G_{n-1}, H_0..H_{n-1}]   (indices into the generator set)
#pragma once
#include "merlin.hpp"

#include <cstdint>
#include <functional>
#include <utility>
#include <vector>

namespace zk {

enum class VarKind : uint8_t { Committed, MulLeft, MulRight, MulOut, One };
struct Var {
  VarKind kind;
  uint32_t idx;
};
inline Var var_one() { return Var{VarKind::One, 0}; }

template <class S>
struct LCt {
  std::vector<std::pair<Var, S>> terms;
  LCt() = default;
  LCt(Var v) { terms.emplace_back(v, S::one()); }   // NOLINT: implicit on purpose
  LCt& add(Var v, const S& c) { terms.emplace_back(v, c); return *this; }
  LCt& sub(Var v, const S& c) { terms.emplace_back(v, -c); return *this; }
};

struct Value {   // spacesuit AllocatedValue: quantity and flavor variables
  Var q, f;
};

// Constraint collection shared by the numeric verifier and the plan builder.
template <class S>
class ConstraintSystemT {
 public:
  using Scalar_t = S;
  using LC = LCt<S>;
  using Deferred = std::function<void(ConstraintSystemT&)>;
  virtual ~ConstraintSystemT() = default;

  void constrain(LC lc) { cons_.push_back(std::move(lc)); }
  // (left, right, out) of a fresh multiplier constrained to the two combinations
  // (virtual: the prover, r1cs_prover.hpp, evaluates the combinations on its witness)
  virtual void multiply(LC left, LC right, Var out[3]) {
    allocate_multiplier(out);
    left.sub(out[0], S::one());
    right.sub(out[1], S::one());
    constrain(std::move(left));
    constrain(std::move(right));
  }
  virtual void allocate_multiplier(Var out[3]) {
    const uint32_t i = num_vars_++;
    out[0] = Var{VarKind::MulLeft, i}; out[1] = Var{VarKind::MulRight, i}; out[2] = Var{VarKind::MulOut, i};
  }
  void specify_randomized_constraints(Deferred f) {
    if (phase2_) f(*this); else deferred_.push_back(std::move(f));
  }
  virtual S challenge_scalar(const char* label) = 0;   // second phase only
  size_t num_multipliers() const { return num_vars_; }
  size_t num_constraints() const { return cons_.size(); }

 protected:
  std::vector<LC> cons_;
  std::vector<Deferred> deferred_;
  uint32_t num_vars_ = 0;
  bool phase2_ = false;
  // returns the number of first-phase multipliers; leaves the system in phase 2
  size_t run_second_phase() {
    const size_t n1 = num_vars_;
    phase2_ = true;
    for (size_t i = 0; i < deferred_.size(); ++i) deferred_[i](*this);
    deferred_.clear();
    return n1;
  }
  bool has_deferred() const { return !deferred_.empty(); }
};

struct VerifierMsm {
  std::vector<uint8_t> dyn_scalars, dyn_points, static_scalars;
  std::vector<uint32_t> static_index;
  size_t padded_n = 0;
};

class R1csVerifier : public ConstraintSystemT<Scalar> {
 public:
  explicit R1csVerifier(const char* label) : tr_(label) { tr_.append_message("dom-sep", (const uint8_t*)"r1cs v1", 7); }

  Var commit(const uint8_t commitment[32]) {
    V_.insert(V_.end(), commitment, commitment + 32);
    tr_.append_point("V", commitment);
    return Var{VarKind::Committed, (uint32_t)(V_.size() / 32 - 1)};
  }
  Scalar challenge_scalar(const char* label) override { return tr_.challenge_scalar(label); }

  // Replays the transcript over `proof` and fills `out`.  gens_capacity = number of (G, H)
  // pairs in the generator set (static index of H_i is 2 + gens_capacity + i).
  // r = verifier's random weight.  false: malformed proof (the reference returns Err).
  bool prepare(const uint8_t* proof, size_t len, const Scalar& r, size_t gens_capacity, VerifierMsm& out) {
    // the one-phase wire format (version byte 0: A_I2, A_O2, S2 left out, 13 + 2k elements) is the two-phase one with
    // the identity in their place (upstream R1CSProof::from_bytes)
    std::vector<uint8_t> expanded;
    if (len >= 1 + 32 * 13 && proof[0] == 0 && (len - 1) % 32 == 0) {
      expanded.assign(len + 96, 0);
      expanded[0] = 1;
      std::memcpy(&expanded[1], proof + 1, 96);
      std::memcpy(&expanded[193], proof + 97, len - 97);
      proof = expanded.data();
      len += 96;
    }
    if (len < 1 + 32 * 16 || proof[0] != 1 || (len - 1) % 32) return false;
    const size_t words = (len - 1) / 32;
    if ((words - 16) % 2) return false;
    const size_t k = (words - 16) / 2;
    if (k >= 32) return false;
    const uint8_t* pt = proof + 1;
    const uint8_t* scb = pt + 32 * 11;
    const uint8_t* lr = scb + 32 * 3;
    const uint8_t* ab = lr + 64 * k;
    Scalar t_x, t_x_bl, e_bl, a, b;
    if (!Scalar::from_canonical(scb, t_x) || !Scalar::from_canonical(scb + 32, t_x_bl) ||
        !Scalar::from_canonical(scb + 64, e_bl) || !Scalar::from_canonical(ab, a) ||
        !Scalar::from_canonical(ab + 32, b))
      return false;
    const size_t m = V_.size() / 32;
    tr_.append_u64("m", m);
    // validate_and_append_point: the identity is rejected
    if (is_identity(pt) || is_identity(pt + 32) || is_identity(pt + 64)) return false;
    tr_.append_point("A_I1", pt);
    tr_.append_point("A_O1", pt + 32);
    tr_.append_point("S1", pt + 64);
    size_t n1 = num_vars_;
    if (!has_deferred()) {
      tr_.append_message("dom-sep", (const uint8_t*)"r1cs-1phase", 11);
    } else {
      tr_.append_message("dom-sep", (const uint8_t*)"r1cs-2phase", 11);
      n1 = run_second_phase();
    }
    const size_t n = num_vars_;
    size_t pn = 1;
    while (pn < n) pn <<= 1;
    if (((size_t)1 << k) != pn || pn > gens_capacity) return false;
    tr_.append_point("A_I2", pt + 96);
    tr_.append_point("A_O2", pt + 128);
    tr_.append_point("S2", pt + 160);
    const Scalar y = tr_.challenge_scalar("y");
    const Scalar z = tr_.challenge_scalar("z");
    for (int i = 6; i < 11; ++i) if (is_identity(pt + 32 * i)) return false;
    tr_.append_point("T_1", pt + 192);
    tr_.append_point("T_3", pt + 224);
    tr_.append_point("T_4", pt + 256);
    tr_.append_point("T_5", pt + 288);
    tr_.append_point("T_6", pt + 320);
    const Scalar u = tr_.challenge_scalar("u");
    const Scalar x = tr_.challenge_scalar("x");
    tr_.append_scalar("t_x", t_x);
    tr_.append_scalar("t_x_blinding", t_x_bl);
    tr_.append_scalar("e_blinding", e_bl);
    const Scalar w = tr_.challenge_scalar("w");

    // flattened constraints: wL, wR, wO, wV, wc weighted by z, z^2, ...
    std::vector<Scalar> wL(n, Scalar::zero()), wR(n, Scalar::zero()), wO(n, Scalar::zero()), wV(m, Scalar::zero());
    Scalar wc = Scalar::zero(), exp_z = z;
    for (const LC& lc : cons_) {
      for (const auto& term : lc.terms) {
        const Scalar t = exp_z * term.second;
        switch (term.first.kind) {
          case VarKind::MulLeft: wL[term.first.idx] += t; break;
          case VarKind::MulRight: wR[term.first.idx] += t; break;
          case VarKind::MulOut: wO[term.first.idx] += t; break;
          case VarKind::Committed: wV[term.first.idx] -= t; break;
          case VarKind::One: wc -= t; break;
        }
      }
      exp_z *= z;
    }
    // inner-product argument: challenges, their inverses (one inversion), s vector
    tr_.append_message("dom-sep", (const uint8_t*)"ipp v1", 6);
    tr_.append_u64("n", pn);
    std::vector<Scalar> ch(k), ch_inv(k);
    for (size_t j = 0; j < k; ++j) {
      if (is_identity(lr + 64 * j) || is_identity(lr + 64 * j + 32)) return false;
      tr_.append_point("L", lr + 64 * j);
      tr_.append_point("R", lr + 64 * j + 32);
      ch[j] = tr_.challenge_scalar("u");
    }
    Scalar allinv = Scalar::one();
    if (k) {   // Montgomery's trick: one inversion for all challenges
      std::vector<Scalar> prefix(k);
      Scalar acc = Scalar::one();
      for (size_t j = 0; j < k; ++j) { prefix[j] = acc; acc *= ch[j]; }
      Scalar inv = acc.invert();
      allinv = inv;
      for (size_t j = k; j-- > 0;) { ch_inv[j] = inv * prefix[j]; inv *= ch[j]; }
    }
    std::vector<Scalar> u_sq(k), u_inv_sq(k);
    for (size_t j = 0; j < k; ++j) { u_sq[j] = ch[j] * ch[j]; u_inv_sq[j] = ch_inv[j] * ch_inv[j]; }
    std::vector<Scalar> s(pn);
    s[0] = allinv;
    for (size_t i = 1; i < pn; ++i) {
      size_t lg_i = 0;
      while (((size_t)2 << lg_i) <= i) ++lg_i;
      s[i] = s[i - ((size_t)1 << lg_i)] * u_sq[(k - 1) - lg_i];
    }
    const Scalar y_inv = y.invert();
    std::vector<Scalar> yinv_pow(pn), yneg_wR(pn, Scalar::zero());
    yinv_pow[0] = Scalar::one();
    for (size_t i = 1; i < pn; ++i) yinv_pow[i] = yinv_pow[i - 1] * y_inv;
    Scalar delta = Scalar::zero();
    for (size_t i = 0; i < n; ++i) { yneg_wR[i] = wR[i] * yinv_pow[i]; delta += yneg_wR[i] * wL[i]; }

    const Scalar xx = x * x, xxx = xx * x, rxx = r * xx;
    out.padded_n = pn;
    out.dyn_scalars.clear(); out.dyn_points.clear(); out.static_scalars.clear(); out.static_index.clear();
    auto push_dyn = [&](const Scalar& sc, const uint8_t* p) {
      uint8_t bts[32];
      sc.to_bytes(bts);
      out.dyn_scalars.insert(out.dyn_scalars.end(), bts, bts + 32);
      out.dyn_points.insert(out.dyn_points.end(), p, p + 32);
    };
    auto push_static = [&](const Scalar& sc, uint32_t idx) {
      uint8_t bts[32];
      sc.to_bytes(bts);
      out.static_scalars.insert(out.static_scalars.end(), bts, bts + 32);
      out.static_index.push_back(idx);
    };
    push_dyn(x, pt); push_dyn(xx, pt + 32); push_dyn(xxx, pt + 64);
    push_dyn(u * x, pt + 96); push_dyn(u * xx, pt + 128); push_dyn(u * xxx, pt + 160);
    for (size_t i = 0; i < m; ++i) push_dyn(wV[i] * rxx, V_.data() + 32 * i);
    const Scalar rx = r * x, rx3 = rxx * x, rx4 = rxx * xx, rx5 = rxx * xxx, rx6 = rx4 * xx;
    push_dyn(rx, pt + 192); push_dyn(rx3, pt + 224); push_dyn(rx4, pt + 256); push_dyn(rx5, pt + 288);
    push_dyn(rx6, pt + 320);
    for (size_t j = 0; j < k; ++j) push_dyn(u_sq[j], lr + 64 * j);
    for (size_t j = 0; j < k; ++j) push_dyn(u_inv_sq[j], lr + 64 * j + 32);
    push_static(w * (t_x - a * b) + r * (xx * (wc + delta) - t_x), 0);           // B
    push_static(-(e_bl + r * t_x_bl), 1);                                          // B_blinding
    for (size_t i = 0; i < pn; ++i) {
      Scalar g = x * yneg_wR[i] - a * s[i];
      if (i >= n1) g *= u;
      push_static(g, (uint32_t)(2 + i));
    }
    for (size_t i = 0; i < pn; ++i) {
      const Scalar wl = i < n ? wL[i] : Scalar::zero(), wo = i < n ? wO[i] : Scalar::zero();
      Scalar h = yinv_pow[i] * (x * wl + wo - b * s[pn - 1 - i]) - Scalar::one();
      if (i >= n1) h *= u;
      push_static(h, (uint32_t)(2 + gens_capacity + i));
    }
    return true;
  }

 private:
  Transcript tr_;
  std::vector<uint8_t> V_;

  static bool is_identity(const uint8_t p[32]) {
    uint8_t acc = 0;
    for (int i = 0; i < 32; ++i) acc |= p[i];
    return acc == 0;
  }
};

// ---- spacesuit::cloak, verifier side (generic in the constraint system) ------------------
namespace cloak {

template <class CS>
Value allocate_value(CS& cs) {
  Var o[3];
  cs.allocate_multiplier(o);
  return Value{o[0], o[1]};
}

template <class CS>
Var product_minus_z(CS& cs, const std::vector<Var>& x, const typename CS::Scalar_t& z) {
  using LC = typename CS::LC;
  const size_t k = x.size();
  Var o[3];
  cs.multiply(LC(x[k - 1]).sub(var_one(), z), LC(x[k - 2]).sub(var_one(), z), o);
  for (size_t i = k - 2; i-- > 0;) cs.multiply(LC(o[2]), LC(x[i]).sub(var_one(), z), o);
  return o[2];
}

template <class CS>
void scalar_shuffle(CS& cs, std::vector<Var> x, std::vector<Var> y) {
  using S = typename CS::Scalar_t;
  using LC = typename CS::LC;
  const size_t k = x.size();
  if (k == 0) return;
  if (k == 1) { cs.constrain(LC(y[0]).sub(x[0], S::one())); return; }
  cs.specify_randomized_constraints([x, y](ConstraintSystemT<S>& c) {
    const S z = c.challenge_scalar("shuffle challenge");
    const Var px = product_minus_z(c, x, z);
    const Var py = product_minus_z(c, y, z);
    c.constrain(LC(px).sub(py, S::one()));
  });
}

template <class CS>
void value_shuffle(CS& cs, std::vector<Value> x, std::vector<Value> y) {
  using S = typename CS::Scalar_t;
  using LC = typename CS::LC;
  const size_t k = x.size();
  if (k == 0) return;
  if (k == 1) {
    cs.constrain(LC(x[0].q).sub(y[0].q, S::one()));
    cs.constrain(LC(x[0].f).sub(y[0].f, S::one()));
    return;
  }
  cs.specify_randomized_constraints([x, y](ConstraintSystemT<S>& c) {
    const S w = c.challenge_scalar("k-value shuffle challenge");
    std::vector<Var> xs, ys;
    for (size_t i = 0; i < x.size(); ++i) {
      Var o[3];
      c.multiply(LC(x[i].q).add(x[i].f, w), LC(y[i].q).add(y[i].f, w), o);
      xs.push_back(o[0]);
      ys.push_back(o[1]);
    }
    scalar_shuffle(c, xs, ys);
  });
}

template <class CS>
void padded_shuffle(CS& cs, std::vector<Value> x, std::vector<Value> y) {
  const size_t k = x.size() > y.size() ? x.size() : y.size();
  while (x.size() < k) x.push_back(allocate_value(cs));
  while (y.size() < k) y.push_back(allocate_value(cs));
  value_shuffle(cs, x, y);
}

template <class CS>
void mix(CS& cs, Value A, Value B, Value C, Value D) {
  using S = typename CS::Scalar_t;
  using LC = typename CS::LC;
  cs.specify_randomized_constraints([A, B, C, D](ConstraintSystemT<S>& c) {
    const S w = c.challenge_scalar("mix challenge");
    const S w2 = w * w, w3 = w2 * w, w4 = w3 * w, one = S::one();
    LC l, r;
    l.add(A.q, one).sub(C.q, one).add(A.f, w).sub(C.f, w).add(B.q, w2).sub(D.q, w2).add(B.f, w3).sub(D.f, w3);
    r.add(C.q, one).add(A.f, w4).sub(B.f, w4).add(D.q, w2).sub(A.q, w2).sub(B.q, w2).add(D.f, w3).sub(A.f, w3);
    Var o[3];
    c.multiply(std::move(l), std::move(r), o);
    c.constrain(LC(o[2]));
  });
}

// (grouped, merged) of a k-mix over `vals`
template <class CS>
void k_mix(CS& cs, const std::vector<Value>& vals, std::vector<Value>& grouped, std::vector<Value>& merged) {
  const size_t k = vals.size();
  grouped.clear();
  merged.clear();
  if (k <= 1) { grouped = vals; merged = vals; return; }
  for (size_t i = 0; i < k; ++i) grouped.push_back(allocate_value(cs));
  std::vector<Value> mid;
  for (size_t i = 0; i + 2 < k; ++i) mid.push_back(allocate_value(cs));
  for (size_t i = 0; i < k; ++i) merged.push_back(allocate_value(cs));
  for (size_t i = 0; i + 1 < k; ++i)
    mix(cs, i == 0 ? grouped[0] : mid[i - 1], grouped[i + 1], merged[i], (i + 2 == k) ? merged[k - 1] : mid[i]);
}

template <class CS>
void range_proof(CS& cs, Var v, int nbits) {
  using S = typename CS::Scalar_t;
  using LC = typename CS::LC;
  LC acc(v);
  S exp2 = S::one();
  for (int i = 0; i < nbits; ++i) {
    Var o[3];
    cs.allocate_multiplier(o);
    cs.constrain(LC(o[2]));
    cs.constrain(LC(o[0]).add(o[1], S::one()).sub(var_one(), S::one()));
    acc.sub(o[1], exp2);
    exp2 = exp2 + exp2;
  }
  cs.constrain(std::move(acc));
}

template <class CS>
void gadget(CS& cs, const std::vector<Value>& in, const std::vector<Value>& out) {
  std::vector<Value> merge_in, merge_out, split_out, split_in;
  k_mix(cs, in, merge_in, merge_out);
  k_mix(cs, out, split_out, split_in);
  value_shuffle(cs, in, merge_in);
  padded_shuffle(cs, merge_out, split_in);
  value_shuffle(cs, split_out, out);
  for (const Value& o : out) range_proof(cs, o.q, 64);
}

// committed values of a ZkVM `cloak`: variable 2i = quantity, 2i + 1 = flavor of value i
inline std::vector<Value> committed_values(size_t n) {
  std::vector<Value> v;
  for (size_t i = 0; i < n; ++i)
    v.push_back(Value{Var{VarKind::Committed, (uint32_t)(2 * i)}, Var{VarKind::Committed, (uint32_t)(2 * i + 1)}});
  return v;
}

// The statement of a ZkVM `cloak` over committed values: commitments = (q, f) per value, inputs first.
inline bool prepare_tx(const uint8_t* commitments, size_t n_in, size_t n_out, const uint8_t* proof, size_t proof_len,
                       const Scalar& r, size_t gens_capacity, VerifierMsm& out) {
  R1csVerifier cs("ZkVM.r1cs");
  std::vector<Value> vals;
  for (size_t i = 0; i < n_in + n_out; ++i) {
    Value v;
    v.q = cs.commit(commitments + 64 * i);
    v.f = cs.commit(commitments + 64 * i + 32);
    vals.push_back(v);
  }
  std::vector<Value> in(vals.begin(), vals.begin() + n_in), outv(vals.begin() + n_in, vals.end());
  gadget(cs, in, outv);
  return cs.prepare(proof, proof_len, r, gens_capacity, out);
}

}  // namespace cloak
}  // namespace zk
