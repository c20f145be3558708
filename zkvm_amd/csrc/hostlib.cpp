// hostlib.cpp -- the host-only part of the product (scalar field, Merlin transcript,
// R1CS verifier scalar preparation) as a plain C++ shared library, so that the CPU test
// tier can exercise it without a GPU.  libzkgpu.so compiles the very same headers.
#include "r1cs_verifier.hpp"
#include "cloak_plan.hpp"
#include "curve.hpp"
#include "r1cs_prover.hpp"
#include "transcript_tape.hpp"
#include "keccak_coop.hpp"
#include "prover_plan.hpp"
#include "zkvm_tx.hpp"

#include <array>
#include <map>

#include <cstring>

using namespace zk;

extern "C" {

// op: 0 add, 1 sub, 2 mul, 3 neg(a), 4 invert(a), 5 reduce only; inputs are 64-byte wide values
int zkhost_scalar_op(int op, const uint8_t a64[64], const uint8_t b64[64], uint8_t out[32]) {
  const Scalar a = Scalar::from_wide(a64), b = Scalar::from_wide(b64);
  Scalar r;
  switch (op) {
    case 0: r = a + b; break;
    case 1: r = a - b; break;
    case 2: r = a * b; break;
    case 3: r = -a; break;
    case 4: r = a.invert(); break;
    case 5: r = a; break;
    default: return -1;
  }
  r.to_bytes(out);
  return 0;
}

int zkhost_scalar_is_canonical(const uint8_t b[32]) {
  Scalar s;
  return Scalar::from_canonical(b, s) ? 1 : 0;
}

// Transcript::new(label); n x append_message(labels[i], msgs[i]); challenge_bytes(ch_label, out)
int zkhost_merlin(const char* label, int n, const char* const* labels, const uint8_t* const* msgs, const size_t* lens,
                  const char* ch_label, uint8_t* out, size_t out_len) {
  Transcript t(label);
  for (int i = 0; i < n; ++i) t.append_message(labels[i], msgs[i], lens[i]);
  t.challenge_bytes(ch_label, out, out_len);
  return 0;
}

// cloak::prepare_tx; buffers sized by the caller: dyn 32 * 64 each, static 32 * (2 + 2 * cap), index 4 * (2 + 2 * cap)
int zkhost_cloak_prepare(const uint8_t* commitments, size_t n_in, size_t n_out, const uint8_t* proof, size_t proof_len,
                         const uint8_t r64[64], size_t gens_capacity, uint8_t* dyn_scalars, uint8_t* dyn_points,
                         size_t* n_dyn, uint8_t* static_scalars, uint32_t* static_index, size_t* n_static,
                         size_t* padded_n) {
  VerifierMsm m;
  if (!cloak::prepare_tx(commitments, n_in, n_out, proof, proof_len, Scalar::from_wide(r64), gens_capacity, m)) return 1;
  *n_dyn = m.dyn_scalars.size() / 32;
  *n_static = m.static_scalars.size() / 32;
  *padded_n = m.padded_n;
  std::memcpy(dyn_scalars, m.dyn_scalars.data(), m.dyn_scalars.size());
  std::memcpy(dyn_points, m.dyn_points.data(), m.dyn_points.size());
  std::memcpy(static_scalars, m.static_scalars.data(), m.static_scalars.size());
  std::memcpy(static_index, m.static_index.data(), m.static_index.size() * 4);
  return 0;
}

// A constraint system described as data (include/zkgpu.h, zkgpu_r1cs_desc) through the host verifier:
// same argument conventions as zkgpu_r1cs_plan_create; output as zkhost_cloak_prepare.
int zkhost_r1cs_prepare(const char* label, uint32_t m, uint32_t n1, uint32_t n, uint32_t n_chal, const char* const* chal_labels,
                        uint32_t n_cons, const uint64_t* term_offsets, const uint8_t* kinds, const uint32_t* idx,
                        const uint8_t* coeff, const int32_t* chal, const uint32_t* power, const uint8_t* commitments,
                        const uint8_t* proof, size_t proof_len, const uint8_t r64[64], size_t gens_capacity,
                        uint8_t* dyn_scalars, uint8_t* dyn_points, size_t* n_dyn, uint8_t* static_scalars,
                        uint32_t* static_index, size_t* n_static, size_t* padded_n) {
  R1csDesc d;
  d.label = label; d.m = m; d.n1 = n1; d.n = n;
  for (uint32_t i = 0; i < n_chal; ++i) d.chal_names.push_back(chal_labels[i]);
  for (uint32_t q = 0; q < n_cons; ++q) {
    std::vector<R1csDesc::Term> con;
    for (uint64_t t = term_offsets[q]; t < term_offsets[q + 1]; ++t) {
      Scalar c;
      if (kinds[t] > 4 || !Scalar::from_canonical(coeff + 32 * t, c)) return -1;
      con.push_back(R1csDesc::Term{(VarKind)kinds[t], idx[t], c, chal[t], power[t]});
    }
    d.cons.push_back(std::move(con));
  }
  try { (void)plan_from_desc(d); } catch (const std::exception&) { return -1; }
  VerifierMsm msm;
  if (!prepare_desc(d, commitments, proof, proof_len, Scalar::from_wide(r64), gens_capacity, msm)) return 1;
  *n_dyn = msm.dyn_scalars.size() / 32;
  *n_static = msm.static_scalars.size() / 32;
  *padded_n = msm.padded_n;
  std::memcpy(dyn_scalars, msm.dyn_scalars.data(), msm.dyn_scalars.size());
  std::memcpy(dyn_points, msm.dyn_points.data(), msm.dyn_points.size());
  std::memcpy(static_scalars, msm.static_scalars.data(), msm.static_scalars.size());
  std::memcpy(static_index, msm.static_index.data(), msm.static_index.size() * 4);
  return 0;
}

// Keccak-f[1600] through the emulated wavefront of keccak_coop.hpp (the algorithm k_transcript_coop runs)
// The lazy scalar form of sc_dev.hpp (scl) against the canonical one (scm) on pseudo-random chains of operations:
// returns the number of mismatches (tests/test_host_logic.py expects 0).
uint64_t zkhost_scl_selftest(uint64_t seed, uint32_t rounds) {
  uint64_t st = seed ? seed : 1, bad = 0;
  auto rnd = [&st]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
  auto rnd_scm = [&]() {
    uint32_t w[16];
    for (auto& x : w) x = rnd();
    return scm_from_wide(w);
  };
  auto same = [](const scm& a, const scl& b) {
    uint32_t w[8];
    scl_canon_words(w, b);
    bool ok = true;
    for (int i = 0; i < 8; ++i) ok &= w[i] == a.v[i];
    return ok;
  };
  for (uint32_t r = 0; r < rounds; ++r) {
    scm a = rnd_scm(), b = rnd_scm(), c = rnd_scm();
    if (r == 0) { a = scm_zero(); }
    if (r == 1) { b = scm_neg(scm_one()); c = b; a = b; }
    const scl la = scl_from_scm(a), lb = scl_from_scm(b), lc = scl_from_scm(c);
    if (!same(scm_mul(a, b), scl_mul(la, lb))) ++bad;
    if (!same(scm_add(a, b), scl_add(la, lb))) ++bad;
    if (!same(scm_sub(a, b), scl_sub(la, lb))) ++bad;
    if (!same(scm_neg(a), scl_neg(la))) ++bad;
    // (a - b) (c - a) + sum of 16 products - b, everything lazy
    scm acc = scm_mul(scm_sub(a, b), scm_sub(c, a));
    scl lacc = scl_mul(scl_sub(la, lb), scl_sub(lc, la));
    scm x = a; scl lx = la;
    for (int i = 0; i < 16; ++i) {
      x = scm_mul(x, c); lx = scl_mul(lx, lc);
      const bool neg = rnd() & 1;
      acc = neg ? scm_sub(acc, x) : scm_add(acc, x);
      lacc = scl_add(lacc, scl_cneg(lx, neg));
    }
    acc = scm_sub(acc, b); lacc = scl_sub(lacc, lb);
    if (!same(acc, lacc)) ++bad;
    if (!same(acc, scl_weak(lacc))) ++bad;
    if (!same(scm_mul(acc, a), scl_mul(lacc, la))) ++bad;
    // the Euclidean inverse of the device prover against Fermat's
    {
      const scm inv_f = scm_invert(a), inv_e = pv_invert(a), inv_u = pv_invert_uniform(a);
      for (int i = 0; i < 8; ++i) if (inv_f.v[i] != inv_e.v[i]) { ++bad; break; }
      for (int i = 0; i < 8; ++i) if (inv_f.v[i] != inv_u.v[i]) { ++bad; break; }
    }
    // plain <-> Montgomery
    uint32_t plain[8];
    scm_to_words(plain, a);
    const scl lp = scl_mul(la, scl_plain_one());
    uint32_t w[8];
    scl_canon_words(w, lp);
    for (int i = 0; i < 8; ++i) if (w[i] != plain[i]) { ++bad; break; }
    if (!same(a, scl_mul(lp, scl_r2()))) ++bad;
    if (!same(scm_one(), scl_one())) ++bad;
    // wave-tree shape: 64 values added with a carry after each doubling
    scl tree = lx; scm stree = x;
    for (int i = 0; i < 6; ++i) { tree = scl_add_c(tree, tree); stree = scm_add(stree, stree); }
    if (!same(stree, tree)) ++bad;
  }
  return bad;
}

void zkhost_keccak_coop(uint64_t state[25]) { coop::keccak_f1600_emulated(state); }

// The cooperative form of the transcript on the host: the tape regrouped into segments
// (build_coop_segments), each word's absorbed bytes gathered as k_tape_gather does, the state spread over
// an emulated wavefront as in k_transcript_coop.  out: 32-byte challenge scalars in the order of
// zkhost_tape_challenges; returns their count or -1.
int zkhost_coop_challenges(uint32_t n_in, uint32_t n_out, const uint8_t* commitments, const uint8_t* proof,
                           size_t proof_len, uint8_t* out, size_t capacity) {
  const CloakPlan plan = PlanBuilder::build(n_in, n_out);
  const uint32_t m = plan.m, k = plan.k, n_chal2 = (uint32_t)plan.chal_names.size();
  if (proof_len != 1 + 32ull * (16 + 2 * k)) return -1;
  const uint32_t ch_fixed = 14;
  Transcript tr("ZkVM.r1cs");
  tr.append_message("dom-sep", (const uint8_t*)"r1cs v1", 7);
  uint32_t init[52];
  tr.export_state(init);
  const std::vector<uint32_t> tape = build_r1cs_verifier_tape(init[50], init[51], m, plan.chal_names, k, plan.pn, ch_fixed);
  const CoopSegments segs = build_coop_segments(tape, m);
  if (!segs.n_seg()) return -1;
  using KC = coop::KeccakCoop<coop::HostTraits>;
  const auto consts = coop::host_consts();
  coop::LaneVec lo, hi;
  for (uint32_t i = 0; i < 64; ++i) {
    const coop::KcLane kl = coop::kc_lane(i);
    lo.l[i] = kl.live ? init[2 * kl.q] : 0;
    hi.l[i] = kl.live ? init[2 * kl.q + 1] : 0;
  }
  std::map<uint32_t, std::vector<uint8_t>> got;
  for (uint32_t sgm = 0; sgm < segs.n_seg(); ++sgm) {
    const uint32_t info = segs.info[sgm], slot = info & 0xffffu;
    if (slot) {
      std::vector<uint8_t> bytes(64);
      for (uint32_t i = 0; i < 64; ++i) {
        const coop::KcLane kl = coop::kc_lane(i);
        if (kl.primary && kl.q < 8) { std::memcpy(&bytes[8 * kl.q], &lo.l[i], 4); std::memcpy(&bytes[8 * kl.q + 4], &hi.l[i], 4); }
        if (kl.live && kl.q < 8) { lo.l[i] = 0; hi.l[i] = 0; }
      }
      got[slot - 1] = bytes;
    }
    for (uint32_t i = 0; i < 64; ++i) {
      const coop::KcLane kl = coop::kc_lane(i);
      if (!kl.live) continue;
      uint64_t w = (uint64_t)segs.consts[sgm * 50 + 2 * kl.q] | ((uint64_t)segs.consts[sgm * 50 + 2 * kl.q + 1] << 32);
      for (int b = 0; b < 8; ++b) {
        const uint32_t idx = segs.map[sgm * 200 + 8 * kl.q + b];
        if (idx) { const uint32_t j = idx - 1; w ^= (uint64_t)(j < 32 * m ? commitments[j] : proof[1 + j - 32 * m]) << (8 * b); }
      }
      lo.l[i] ^= (uint32_t)w; hi.l[i] ^= (uint32_t)(w >> 32);
    }
    if (info >> 31) KC::permute(lo, hi, consts);
  }
  std::vector<uint32_t> order = {0, 1, 2, 3, 4};
  for (uint32_t j = 0; j < n_chal2; ++j) order.push_back(ch_fixed + j);
  for (uint32_t j = 0; j < k; ++j) order.push_back(ch_fixed + n_chal2 + j);
  if (order.size() > capacity) return -1;
  for (size_t i = 0; i < order.size(); ++i) {
    if (!got.count(order[i])) return -1;
    Scalar::from_wide(got[order[i]].data()).to_bytes(out + 32 * i);
  }
  return (int)order.size();
}

// The device-side transcript tape (transcript_tape.hpp) against the Transcript class on the same
// proof: out_tape / out_direct receive 32-byte challenge scalars in slot order y z u x w, the
// second-phase challenges, the k inner-product challenges.  Returns their count, or -1.
int zkhost_tape_challenges(uint32_t n_in, uint32_t n_out, const uint8_t* commitments, const uint8_t* proof,
                           size_t proof_len, uint8_t* out_tape, uint8_t* out_direct, size_t capacity) {
  const CloakPlan plan = PlanBuilder::build(n_in, n_out);
  const uint32_t m = plan.m, k = plan.k, n_chal2 = (uint32_t)plan.chal_names.size();
  if (proof_len != 1 + 32ull * (16 + 2 * k)) return -1;
  const uint32_t ch_fixed = 14;
  Transcript tr("ZkVM.r1cs");
  tr.append_message("dom-sep", (const uint8_t*)"r1cs v1", 7);
  uint32_t init[52];
  tr.export_state(init);
  const std::vector<uint32_t> tape = build_r1cs_verifier_tape(init[50], init[51], m, plan.chal_names, k, plan.pn, ch_fixed);
  uint8_t state[200];
  std::memcpy(state, init, 200);
  std::map<uint32_t, std::vector<uint8_t>> got;
  run_tape_host(tape, state, commitments, proof + 1, [](uint8_t* st) {
    uint64_t a[25];
    std::memcpy(a, st, 200);
    keccak_f1600(a);
    std::memcpy(st, a, 200);
  }, got);
  std::vector<uint32_t> order = {0, 1, 2, 3, 4};
  for (uint32_t j = 0; j < n_chal2; ++j) order.push_back(ch_fixed + j);
  for (uint32_t j = 0; j < k; ++j) order.push_back(ch_fixed + n_chal2 + j);
  if (order.size() > capacity) return -1;
  for (size_t i = 0; i < order.size(); ++i) {
    if (!got.count(order[i])) return -1;
    Scalar::from_wide(got[order[i]].data()).to_bytes(out_tape + 32 * i);
  }
  // the same sequence through the Transcript class
  const uint8_t* f = proof + 1;
  std::vector<Scalar> direct(order.size());
  for (uint32_t i = 0; i < m; ++i) tr.append_message("V", commitments + 32 * i, 32);
  {
    uint8_t b[8] = {0};
    for (int q = 0; q < 8; ++q) b[q] = (uint8_t)((uint64_t)m >> (8 * q));
    tr.append_message("m", b, 8);
  }
  tr.append_message("A_I1", f, 32); tr.append_message("A_O1", f + 32, 32); tr.append_message("S1", f + 64, 32);
  if (n_chal2 == 0) {
    tr.append_message("dom-sep", (const uint8_t*)"r1cs-1phase", 11);
  } else {
    tr.append_message("dom-sep", (const uint8_t*)"r1cs-2phase", 11);
    for (uint32_t j = 0; j < n_chal2; ++j) {
      direct[5 + j] = tr.challenge_scalar(plan.chal_names[j].c_str());
    }
  }
  tr.append_message("A_I2", f + 96, 32); tr.append_message("A_O2", f + 128, 32); tr.append_message("S2", f + 160, 32);
  direct[0] = tr.challenge_scalar("y");
  direct[1] = tr.challenge_scalar("z");
  const char* tl[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
  for (int i = 0; i < 5; ++i) tr.append_message(tl[i], f + 32 * (6 + i), 32);
  direct[2] = tr.challenge_scalar("u");
  direct[3] = tr.challenge_scalar("x");
  tr.append_message("t_x", f + 32 * 11, 32);
  tr.append_message("t_x_blinding", f + 32 * 12, 32);
  tr.append_message("e_blinding", f + 32 * 13, 32);
  direct[4] = tr.challenge_scalar("w");
  tr.append_message("dom-sep", (const uint8_t*)"ipp v1", 6);
  {
    uint8_t b[8] = {0};
    for (int q = 0; q < 8; ++q) b[q] = (uint8_t)((uint64_t)plan.pn >> (8 * q));
    tr.append_message("n", b, 8);
  }
  for (uint32_t j = 0; j < k; ++j) {
    tr.append_message("L", f + 32 * (14 + 2 * j), 32);
    tr.append_message("R", f + 32 * (14 + 2 * j + 1), 32);
    direct[5 + n_chal2 + j] = tr.challenge_scalar("u");
  }
  for (size_t i = 0; i < order.size(); ++i) direct[i].to_bytes(out_direct + 32 * i);
  return (int)order.size();
}

}  // extern "C"

namespace {
// Reference evaluation of prover rows on the host (CPU tests only; the product evaluates them with
// zkgpu_msm_ps_batch): bucket method, window 8, over decoded generator points.
void host_rows(const std::vector<ge>& gens, const std::vector<MsmRow>& rows, std::vector<uint8_t>& out) {
  out.assign(32 * rows.size(), 0);
  for (size_t r = 0; r < rows.size(); ++r) {
    const MsmRow& row = rows[r];
    ge acc;
    ge_identity(acc);
    std::vector<std::array<uint8_t, 32>> sb(row.scalars.size());
    for (size_t i = 0; i < sb.size(); ++i) row.scalars[i].to_bytes(sb[i].data());
    for (int win = 31; win >= 0; --win) {
      for (int d = 0; d < 8; ++d) ge_double(acc, acc);
      std::vector<ge> bucket(256);
      std::vector<char> used(256, 0);
      for (size_t i = 0; i < sb.size(); ++i) {
        const unsigned b = sb[i][win];
        if (!b) continue;
        if (used[b]) ge_add(bucket[b], bucket[b], gens[row.index[i]]);
        else { bucket[b] = gens[row.index[i]]; used[b] = 1; }
      }
      ge run, sum;
      ge_identity(run);
      ge_identity(sum);
      for (int b = 255; b >= 1; --b) {
        if (used[b]) ge_add(run, run, bucket[b]);
        ge_add(sum, sum, run);
      }
      ge_add(acc, acc, sum);
    }
    uint32_t enc[8];
    ristretto_encode(enc, acc);
    std::memcpy(&out[32 * r], enc, 32);
  }
}
}  // namespace

extern "C" {

// One cloak proof on the host: the product's prover (r1cs_prover.hpp) with its rows evaluated by the
// reference MSM above.  generators = [B, B_blinding, G_0..G_{cap-1}, H_0..H_{cap-1}] compressed.
// Returns 0; commitments = 64 (n_in + n_out) bytes, proof_len_out = bytes written to proof.
int zkhost_cloak_prove(uint32_t n_in, uint32_t n_out, const uint64_t* quantities, const uint8_t* flavors,
                       const uint8_t seed[32], const uint8_t* generators, size_t gens_capacity, uint8_t* commitments,
                       uint8_t* proof, size_t proof_cap, size_t* proof_len_out, size_t* multipliers) {
  std::vector<ge> gens(2 + 2 * gens_capacity);
  for (size_t i = 0; i < gens.size(); ++i) {
    uint32_t w[8];
    std::memcpy(w, generators + 32 * i, 32);
    if (!ristretto_decode(gens[i], w)) return -1;
  }
  std::unique_ptr<R1csProver> prp = cloak_prover(n_in, n_out, quantities, flavors, seed, gens_capacity);
  R1csProver& pr = *prp;
  std::vector<MsmRow> rows;
  std::vector<uint8_t> pts;
  pr.begin(rows);
  while (!pr.done()) {
    host_rows(gens, rows, pts);
    pr.step(pts.data(), rows);
  }
  if (pr.failed() || pr.proof().size() > proof_cap) return -2;
  std::memcpy(commitments, pr.commitments().data(), pr.commitments().size());
  std::memcpy(proof, pr.proof().data(), pr.proof().size());
  *proof_len_out = pr.proof().size();
  if (multipliers) *multipliers = pr.multipliers();
  return 0;
}

// The prover for a constraint system described as data (desc_prover) with the reference MSM of this library:
// description arrays as zkhost_r1cs_prepare; mult_def: 2 per multiplier; values: m x 32 bytes; given: n_given x 64
// bytes (left, right); blindings derived from the seed as oracle/gadgets.c does ("blinding", i).
int zkhost_r1cs_prove(const char* label, uint32_t m, uint32_t n1, uint32_t n, uint32_t n_chal, const char* const* chal_labels,
                      uint32_t n_cons, const uint64_t* term_offsets, const uint8_t* kinds, const uint32_t* idx,
                      const uint8_t* coeff, const int32_t* chal, const uint32_t* power, const uint32_t* mult_def,
                      const uint8_t* values, const uint8_t* given, size_t n_given, const uint8_t seed[32],
                      const uint8_t* generators, size_t gens_capacity, uint8_t* commitments, uint8_t* proof, size_t proof_cap,
                      size_t* proof_len_out) {
  R1csDesc d;
  d.label = label; d.m = m; d.n1 = n1; d.n = n;
  for (uint32_t i = 0; i < n_chal; ++i) d.chal_names.push_back(chal_labels[i]);
  for (uint32_t q = 0; q < n_cons; ++q) {
    std::vector<R1csDesc::Term> con;
    for (uint64_t t = term_offsets[q]; t < term_offsets[q + 1]; ++t) {
      Scalar c;
      if (kinds[t] > 4 || !Scalar::from_canonical(coeff + 32 * t, c)) return -1;
      con.push_back(R1csDesc::Term{(VarKind)kinds[t], idx[t], c, chal[t], power[t]});
    }
    d.cons.push_back(std::move(con));
  }
  std::vector<ge> gens(2 + 2 * gens_capacity);
  for (size_t i = 0; i < gens.size(); ++i) {
    uint32_t w[8];
    std::memcpy(w, generators + 32 * i, 32);
    if (!ristretto_decode(gens[i], w)) return -1;
  }
  std::vector<Scalar> vals, bl;
  for (uint32_t i = 0; i < m; ++i) {
    uint8_t wide[64] = {0};
    std::memcpy(wide, values + 32 * i, 32);
    vals.push_back(Scalar::from_wide(wide));
    bl.push_back(R1csProver::derive_scalar(seed, "blinding", i));
  }
  std::vector<std::pair<Scalar, Scalar>> gv;
  for (size_t i = 0; i < n_given; ++i) {
    uint8_t wl[64] = {0}, wr[64] = {0};
    std::memcpy(wl, given + 64 * i, 32); std::memcpy(wr, given + 64 * i + 32, 32);
    gv.emplace_back(Scalar::from_wide(wl), Scalar::from_wide(wr));
  }
  std::vector<uint32_t> md(mult_def, mult_def + 2 * (size_t)n);
  std::unique_ptr<R1csProver> prp = desc_prover(d, md, vals, bl, gv, seed, gens_capacity);
  R1csProver& pr = *prp;
  std::vector<MsmRow> rows;
  std::vector<uint8_t> pts;
  pr.begin(rows);
  while (!pr.done()) {
    host_rows(gens, rows, pts);
    pr.step(pts.data(), rows);
  }
  if (pr.failed() || pr.proof().size() > proof_cap) return -2;
  std::memcpy(commitments, pr.commitments().data(), pr.commitments().size());
  std::memcpy(proof, pr.proof().data(), pr.proof().size());
  *proof_len_out = pr.proof().size();
  return 0;
}

}  // extern "C"

// ---- the DEVICE prover (prover_dev.hpp) run on the host with one "thread" per proof: the same phase functions the
// kernels call, the reference multiscalar multiplication of this library between them.  CPU tests compare the bytes
// with the host prover's and the oracle's.
namespace {
// pv_emulate's way through a phase: 0 the whole phase in one call (PV_ALL), 1 stage by stage in the device's order, the
// one-thread stages on an environment that inverts as the lane kernels do (zkhost_set_pv_staged; the CPU tests run both)
int g_pv_staged = 0;
struct PvHostEnv {
  static constexpr bool kInvertInEveryLane = false;
  uint32_t st[52];
  uint32_t tid() const { return 0; }
  uint32_t nt() const { return 1; }
  void sync() {}
  void sum(scl*, int) {}
  uint32_t* strobe() { return st; }
};

struct PvHostLaneEnv {          // what k_pv_lanes / k_pv_ipa_lanes give a proof: one thread, the fixed-chain inverse
  static constexpr bool kInvertInEveryLane = true;
  uint32_t st[52];
  uint32_t tid() const { return 0; }
  uint32_t nt() const { return 1; }
  void sync() {}
  void sum(scl*, int) {}
  uint32_t* strobe() { return st; }
};

void words_of(const uint8_t* b, size_t n_words, std::vector<uint32_t>& out) {
  for (size_t i = 0; i < n_words; ++i) out.push_back((uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24));
}
Scalar scalar_of_words(const uint32_t* w) {
  uint8_t wide[64] = {0};
  for (int i = 0; i < 8; ++i) for (int b = 0; b < 4; ++b) wide[4 * i + b] = (uint8_t)(w[i] >> (8 * b));
  return Scalar::from_wide(wide);
}
void rows_points(const std::vector<ge>& gens, const uint32_t* scalars, const PvRows& lay, std::vector<uint32_t>& out_words) {
  std::vector<MsmRow> rows(lay.offsets.size() - 1);
  for (size_t r = 0; r + 1 < lay.offsets.size(); ++r)
    for (uint64_t k = lay.offsets[r]; k < lay.offsets[r + 1]; ++k) rows[r].add(scalar_of_words(scalars + 8 * k), lay.index[k]);
  std::vector<uint8_t> pts;
  host_rows(gens, rows, pts);
  out_words.clear();
  words_of(pts.data(), pts.size() / 4, out_words);
}

int pv_emulate(const R1csDesc& d, const std::vector<uint32_t>& mult_def, const std::vector<uint32_t>& values, const std::vector<uint32_t>& blindings,
               const std::vector<uint32_t>& given, const uint8_t seed[32], const uint8_t* generators, size_t cap, uint8_t* commitments,
               uint8_t* proof, size_t proof_cap, size_t* proof_len_out) {
  std::vector<ge> gens(2 + 2 * cap);
  for (size_t i = 0; i < gens.size(); ++i) {
    uint32_t w[8];
    std::memcpy(w, generators + 32 * i, 32);
    if (!ristretto_decode(gens[i], w)) return -1;
  }
  PvHostPlan hp;
  try { hp = pv_build(d, mult_def, cap); } catch (const std::exception&) { return -3; }
  const PvShape& sh = hp.sh;
  if (given.size() != 16 * (size_t)sh.n_given || values.size() != 8 * (size_t)sh.m || sh.proof_len > proof_cap) return -4;
  const PvPlan P = hp.view();
  std::vector<uint32_t> state(sh.state_words, 0), rows0(16 * std::max<uint32_t>(sh.m, 1)), rows1(8 * std::max<uint32_t>(sh.r1_terms, 1)), rows2(8 * std::max<uint32_t>(sh.r2_terms, 1)),
      rows3(5 * 16), lv(8 * sh.pn), rv(8 * sh.pn), cg(8 * sh.pn), ch(8 * sh.pn), w(8), uu(16), rng_seed;
  uint8_t rs[32];
  R1csProver::derive(seed, "rng", 0, rs, 32);
  words_of(rs, 8, rng_seed);
  std::vector<uint8_t> pbytes(sh.proof_stride, 0);
  PvBatch B{};
  B.state = state.data(); B.values = values.data(); B.blindings = blindings.data(); B.given = given.empty() ? values.data() : given.data();
  B.rng_seed = rng_seed.data(); B.proofs = pbytes.data(); B.rows0 = rows0.data(); B.rows1 = rows1.data(); B.rows2 = rows2.data(); B.rows3 = rows3.data();
  B.ipa_lv = lv.data(); B.ipa_rv = rv.data(); B.ipa_cg = cg.data(); B.ipa_ch = ch.data(); B.ipa_w = w.data(); B.ipa_u = uu.data();
  PvHostEnv env;
  PvHostLaneEnv lane;
  const bool staged = g_pv_staged != 0;
  std::vector<uint32_t> pts;
  pv_phase0(env, sh, B, 0);
  rows_points(gens, rows0.data(), pv_rows_pairs(sh.m), pts);
  for (size_t i = 0; i < 8 * (size_t)sh.m; ++i) for (int b = 0; b < 4; ++b) commitments[4 * i + b] = (uint8_t)(pts[i] >> (8 * b));
  if (staged) {
    pv_phase1(lane, sh, P, B, 0, pts.data(), 1); pv_phase1(env, sh, P, B, 0, pts.data(), 2);
    pv_phase1(lane, sh, P, B, 0, pts.data(), 4); pv_phase1(env, sh, P, B, 0, pts.data(), 8);
  } else {
    pv_phase1(env, sh, P, B, 0, pts.data());
  }
  pv_rng_draw(sh, state.data(), rows1.data(), 0, sh.n1, PV_IBL1);
  rows_points(gens, rows1.data(), pv_rows_commit(1, 0, sh.n1, cap, false), pts);
  if (staged) {
    pv_phase2(lane, sh, P, B, 0, pts.data(), 1);
    if (sh.n > sh.n1) { pv_phase2(env, sh, P, B, 0, pts.data(), 2); pv_phase2(lane, sh, P, B, 0, pts.data(), 4); pv_phase2(env, sh, P, B, 0, pts.data(), 8); }
  } else {
    pv_phase2(env, sh, P, B, 0, pts.data());
  }
  if (sh.n > sh.n1) pv_rng_draw(sh, state.data(), rows2.data(), sh.n1, sh.n, PV_IBL2);
  rows_points(gens, rows2.data(), pv_rows_commit(1, sh.n1, sh.n, cap, true), pts);
  if (staged) { pv_phase3(lane, sh, P, B, 0, pts.data(), 1); pv_phase3(env, sh, P, B, 0, pts.data(), 2); pv_phase3(lane, sh, P, B, 0, pts.data(), 4); }
  else pv_phase3(env, sh, P, B, 0, pts.data());
  rows_points(gens, rows3.data(), pv_rows_pairs(5), pts);
  if (staged) { pv_phase4(lane, sh, P, B, 0, pts.data(), 1); pv_phase4(env, sh, P, B, 0, pts.data(), 2); }
  else pv_phase4(env, sh, P, B, 0, pts.data());
  if (state[sh.o_flag]) return -2 - (int)state[sh.o_flag];
  // the inner-product rounds (on the device: k_ipa_round + the tables); here with the host's scalars
  std::vector<Scalar> L(sh.pn), R(sh.pn), G(sh.pn), H(sh.pn);
  for (uint32_t i = 0; i < sh.pn; ++i) {
    uint32_t t[8];
    scm a; for (int q = 0; q < 8; ++q) a.v[q] = lv[8 * i + q];
    scm_to_words(t, a); L[i] = scalar_of_words(t);
    for (int q = 0; q < 8; ++q) a.v[q] = rv[8 * i + q];
    scm_to_words(t, a); R[i] = scalar_of_words(t);
    G[i] = scalar_of_words(&cg[8 * i]); H[i] = scalar_of_words(&ch[8 * i]);
  }
  const Scalar wq = scalar_of_words(w.data());
  size_t len = sh.pn;
  for (uint32_t round = 0; round < sh.k; ++round) {
    const size_t half = len / 2;
    Scalar cL = Scalar::zero(), cR = Scalar::zero();
    for (size_t j = 0; j < half; ++j) { cL += L[j] * R[half + j]; cR += L[half + j] * R[j]; }
    std::vector<MsmRow> rows(2);
    for (size_t idx = 0; idx < sh.pn; ++idx) {
      const size_t j = idx % len;
      const bool hi = j >= half;
      const size_t jj = hi ? j - half : j;
      if (hi) { rows[0].add(L[jj] * G[idx], (uint32_t)(2 + idx)); rows[1].add(R[jj] * H[idx], (uint32_t)(2 + cap + idx)); }
      else { rows[0].add(R[half + jj] * H[idx], (uint32_t)(2 + cap + idx)); rows[1].add(L[half + jj] * G[idx], (uint32_t)(2 + idx)); }
    }
    rows[0].add(cL * wq, 0);
    rows[1].add(cR * wq, 0);
    std::vector<uint8_t> lr;
    host_rows(gens, rows, lr);
    std::vector<uint32_t> lrw;
    words_of(lr.data(), 16, lrw);
    if (staged) pv_ipa_round(lane, sh, B, 0, round, lrw.data()); else pv_ipa_round(env, sh, B, 0, round, lrw.data());
    const Scalar u = scalar_of_words(uu.data()), ui = scalar_of_words(uu.data() + 8);
    for (size_t j = 0; j < half; ++j) {
      L[j] = L[j] * u + L[half + j] * ui;
      R[j] = R[j] * ui + R[half + j] * u;
    }
    for (size_t idx = 0; idx < sh.pn; ++idx) {
      const bool hi = (idx % len) >= half;
      G[idx] *= hi ? u : ui;
      H[idx] *= hi ? ui : u;
    }
    len = half;
  }
  uint8_t ab[64];
  L[0].to_bytes(ab); R[0].to_bytes(ab + 32);
  std::vector<uint32_t> abw;
  words_of(ab, 16, abw);
  pv_finish(sh, B, 0, abw.data());
  std::memcpy(proof, pbytes.data(), sh.proof_len);
  *proof_len_out = sh.proof_len;
  return 0;
}
}  // namespace

extern "C" {
void zkhost_set_pv_staged(int on) { g_pv_staged = on; }
int zkhost_prove_dev_cloak(uint32_t n_in, uint32_t n_out, const uint64_t* quantities, const uint8_t* flavors, const uint8_t seed[32],
                           const uint8_t* generators, size_t gens_capacity, uint8_t* commitments, uint8_t* proof, size_t proof_cap,
                           size_t* proof_len_out) {
  R1csDesc d;
  std::vector<uint32_t> md, values, blindings, given;
  PvCloakTrace::trace(n_in, n_out, d, md);
  for (size_t i = 0; i < (size_t)n_in + n_out; ++i) {
    uint8_t b[32];
    Scalar::from_u64(quantities[i]).to_bytes(b); words_of(b, 8, values);
    words_of(flavors + 32 * i, 8, values);
    R1csProver::derive_scalar(seed, "q_blinding", i).to_bytes(b); words_of(b, 8, blindings);
    R1csProver::derive_scalar(seed, "f_blinding", i).to_bytes(b); words_of(b, 8, blindings);
  }
  pv_cloak_given(n_in, n_out, quantities, flavors, given);
  return pv_emulate(d, md, values, blindings, given, seed, generators, gens_capacity, commitments, proof, proof_cap, proof_len_out);
}

int zkhost_prove_dev_r1cs(const char* label, uint32_t m, uint32_t n1, uint32_t n, uint32_t n_chal, const char* const* chal_labels,
                          uint32_t n_cons, const uint64_t* term_offsets, const uint8_t* kinds, const uint32_t* idx,
                          const uint8_t* coeff, const int32_t* chal, const uint32_t* power, const uint32_t* mult_def,
                          const uint8_t* values, const uint8_t* given, size_t n_given, const uint8_t seed[32],
                          const uint8_t* generators, size_t gens_capacity, uint8_t* commitments, uint8_t* proof, size_t proof_cap,
                          size_t* proof_len_out) {
  R1csDesc d;
  d.label = label; d.m = m; d.n1 = n1; d.n = n;
  for (uint32_t i = 0; i < n_chal; ++i) d.chal_names.push_back(chal_labels[i]);
  for (uint32_t q = 0; q < n_cons; ++q) {
    std::vector<R1csDesc::Term> con;
    for (uint64_t t = term_offsets[q]; t < term_offsets[q + 1]; ++t) {
      Scalar c;
      if (kinds[t] > 4 || !Scalar::from_canonical(coeff + 32 * t, c)) return -1;
      con.push_back(R1csDesc::Term{(VarKind)kinds[t], idx[t], c, chal[t], power[t]});
    }
    d.cons.push_back(std::move(con));
  }
  std::vector<uint32_t> md(mult_def, mult_def + 2 * (size_t)n), vw, bw, gw;
  words_of(values, 8 * (size_t)m, vw);
  for (uint32_t i = 0; i < m; ++i) {
    uint8_t b[32];
    R1csProver::derive_scalar(seed, "blinding", i).to_bytes(b);
    words_of(b, 8, bw);
  }
  words_of(given, 16 * n_given, gw);
  return pv_emulate(d, md, vw, bw, gw, seed, generators, gens_capacity, commitments, proof, proof_cap, proof_len_out);
}
}  // extern "C"

// ---- ZkVM transactions (zkvm_tx.hpp): the host half of Tx::verify, with this library's reference group arithmetic for
// the aggregated key.  Out: status (0 ok, 1 invalid, 2 outside the subset), txid, the cloak's shape and commitments,
// the terms of the signature equation (scalars | points, 32 bytes each; *n_sig of them).
extern "C" int zkhost_tx_prepare(const uint8_t* tx, size_t len, uint8_t txid[32], uint32_t* n_in, uint32_t* n_out, uint8_t* commitments,
                                 size_t com_cap, uint8_t* sig_scalars, uint8_t* sig_points, size_t sig_cap, size_t* n_sig,
                                 size_t* proof_offset, size_t* proof_len) {
  using namespace zk::zkvm;
  TxStatement st = tx_prepare(tx, len);
  *n_sig = 0;
  if (st.status != TX_OK) return (int)st.status;
  const size_t n_terms = st.sig_scalars.size() / 32, n_keys = n_terms - 2;
  if (st.commitments.size() > com_cap || n_terms > sig_cap) return -1;
  // aggregated key X = sum a_i X_i with the reference multiscalar multiplication
  std::vector<ge> pts(n_keys);
  MsmRow row;
  for (size_t i = 0; i < n_keys; ++i) {
    uint32_t w[8];
    std::memcpy(w, &st.sig_points[32 * (2 + i)], 32);
    if (!ristretto_decode(pts[i], w)) return (int)TX_INVALID;
    Scalar a;
    Scalar::from_canonical(&st.sig_scalars[32 * (2 + i)], a);
    row.add(a, (uint32_t)i);
  }
  std::vector<uint8_t> agg;
  host_rows(pts, {row}, agg);
  ge B;
  B.X = fe_BASE_X(); B.Y = fe_BASE_Y(); B.Z = fe_one(); B.T = fe_BASE_T();
  uint32_t benc[8];
  ristretto_encode(benc, B);
  tx_finish_signature(st, (const uint8_t*)benc, agg.data());
  std::memcpy(txid, st.txid, 32);
  *n_in = st.n_in; *n_out = st.n_out;
  std::memcpy(commitments, st.commitments.data(), st.commitments.size());
  std::memcpy(sig_scalars, st.sig_scalars.data(), st.sig_scalars.size());
  std::memcpy(sig_points, st.sig_points.data(), st.sig_points.size());
  *n_sig = n_terms;
  *proof_offset = (size_t)(st.proof - tx);
  *proof_len = st.proof_len;
  return 0;
}

// PvRngCoop (the TranscriptRng's draws on a state spread over an emulated wavefront) against PvRng (one lane, state in
// registers) and thereby against the byte-wise STROBE: n draws from a keyed generator; returns mismatches.
extern "C" uint64_t zkhost_rng_coop_selftest(uint64_t seed, uint32_t n_draws) {
  using namespace zk::coop;
  uint32_t w[52];
  {
    Transcript t("rng selftest");
    uint8_t sd[32];
    for (int i = 0; i < 32; ++i) sd[i] = (uint8_t)(seed >> (8 * (i & 7))) ^ (uint8_t)i;
    t.rekey_with_witness("v_blinding", sd, 32);
    t.finalize_rng(sd);
    t.export_state(w);
  }
  PvRng a;
  a.load(w);
  if (!a.fast()) return ~0ull;
  LaneVec lo, hi;
  PvRngCoop<HostTraits>::Masks m;
  for (uint32_t i = 0; i < 64; ++i) {
    const KcLane k = kc_lane(i);
    lo.l[i] = k.live ? w[2 * k.q] : 0;
    hi.l[i] = k.live ? w[2 * k.q + 1] : 0;
    auto holds = [&k](uint32_t q) { return (k.live && k.q == q) ? ~0u : 0u; };
    m.w4.l[i] = holds(4); m.w5.l[i] = holds(5); m.w8.l[i] = holds(8); m.w9.l[i] = holds(9); m.w20.l[i] = holds(20);
    m.keep.l[i] = (k.live && k.q < 8) ? 0u : ~0u;
  }
  const auto c = host_consts();
  uint64_t bad = 0;
  for (uint32_t d = 0; d < n_draws; ++d) {
    const scm want = a.draw();
    PvRngCoop<HostTraits>::draw(lo, hi, c, m, d == 0 && w[50] == 32);
    uint32_t wd[16];
    for (uint32_t i = 0; i < 64; ++i) {
      const KcLane k = kc_lane(i);
      if (k.primary && k.q < 8) { wd[2 * k.q] = lo.l[i]; wd[2 * k.q + 1] = hi.l[i]; }
    }
    PvRngCoop<HostTraits>::taken(lo, hi, m);
    const scm got = scm_from_wide(wd);
    for (int q = 0; q < 8; ++q) if (got.v[q] != want.v[q]) { ++bad; break; }
  }
  uint32_t wa[52];
  a.store(wa);
  for (uint32_t i = 0; i < 64; ++i) {
    const KcLane k = kc_lane(i);
    if (k.primary && (lo.l[i] != wa[2 * k.q] || hi.l[i] != wa[2 * k.q + 1])) ++bad;
  }
  return bad;
}

// HostPool (host_pool.hpp, the workers behind every host stage of the product): every index of every call visited exactly
// once for a range of sizes and thread counts, calls from `callers` threads at once (one at a time gets the pool, the
// others wait their turn), and repeated use of the sleeping workers.  Returns the number of violations.
#include "host_pool.hpp"
// what host_threads = 0 means: the CPUs this process may keep busy (affinity mask, control-group quota)
extern "C" int zkhost_usable_cpus() { return zk::usable_cpus(); }

extern "C" uint64_t zkhost_pool_selftest(uint32_t rounds, uint32_t callers) {
  std::atomic<uint64_t> bad{0}, took{0}, refused{0};
  auto one = [&](uint32_t salt) {
    static const size_t sizes[] = {0, 1, 2, 15, 16, 17, 255, 1000, 4097};
    static const int threads[] = {2, 3, 8, 32, 100};
    for (uint32_t r = 0; r < rounds; ++r)
      for (size_t n : sizes)
        for (int nt : threads) {
          std::vector<std::atomic<uint32_t>> hits(n);
          for (auto& h : hits) h.store(0);
          const std::function<void(size_t)> f = [&](size_t i) { (void)salt; hits[i].fetch_add(1); };
          if (zk::HostPool::get().run(n, std::min(nt, zk::HostPool::MAX_WORKERS + 1), f)) {
            ++took;
            for (auto& h : hits) if (h.load() != 1) ++bad;
          } else {
            ++refused;
            for (auto& h : hits) if (h.load() != 0) ++bad;   // a refused call must not have touched anything
          }
        }
  };
  std::vector<std::thread> th;
  for (uint32_t c = 1; c < callers; ++c) th.emplace_back(one, c);
  one(0);
  for (auto& t : th) t.join();
  if (took.load() == 0) ++bad;
  if (refused.load() != 0) ++bad;                     // only a forked process is refused
  return bad.load();
}

// ---- the framing of the sharded verification's one exchange (comm_frame.hpp), for the CPU tests: a world of any size is
// ---- the test concatenating the slots its "ranks" packed
#include "ticket_cut.hpp"
// the device batches a burst of tickets leaves in (tickets per batch) under the equal-parts policy; returns their number
extern "C" size_t zkhost_ticket_cut(const uint64_t* sizes, size_t n, uint64_t target, uint64_t* out_counts, size_t cap) {
  std::vector<size_t> v(sizes, sizes + n);
  const std::vector<size_t> cut = zk::ticket_cut(v, (size_t)target);
  for (size_t i = 0; i < cut.size() && i < cap; ++i) out_counts[i] = cut[i];
  return cut.size();
}

#include "comm_frame.hpp"
extern "C" size_t zkhost_comm_slot_bytes(const uint64_t* cuts, int world) { return commframe::slot_bytes(cuts, world); }
extern "C" void zkhost_comm_pack(uint8_t* out, size_t slot, const uint64_t* cuts, int rank, const uint8_t* local_bitmap, int local_status) {
  commframe::pack(out, slot, cuts, rank, local_bitmap, local_status);
}
extern "C" int zkhost_comm_unpack(const uint8_t* all, size_t slot, const uint64_t* cuts, int world, int rank, uint8_t* whole) {
  return commframe::unpack(all, slot, cuts, world, rank, whole);
}

// ---- the keys-first pass of zkgpu_tx_verify_batch (tx_prepare_many with only = P_MUSIG: the plan drops every other hash job)
// ---- against the full pass, for the CPU tests: per transaction status and a SHA3-512 digest over what BOTH must leave in
// ---- the statement before the signature challenge -- arity, proof length, commitments, s, -1, the MuSig coefficients
// ---- a_i, R and the keys X_i.
extern "C" void zkhost_tx_rows_group(const uint8_t* txs, const uint64_t* offs, size_t count, int lockstep, int keys_only,
                                     uint8_t* status, uint8_t* digest /*64 per tx*/) {
  using namespace zk::zkvm;
  for (size_t g = 0; g < count; g += 8) {
    const size_t n = std::min<size_t>(8, count - g);
    const uint8_t* p[8]; size_t l[8];
    TxStatement st[8];
    for (size_t i = 0; i < n; ++i) { p[i] = txs + offs[g + i]; l[i] = (size_t)(offs[g + i + 1] - offs[g + i]); }
    tx_prepare_many(p, l, st, n, lockstep != 0, keys_only ? (uint8_t)P_MUSIG : ALL_PROTOS);
    for (size_t i = 0; i < n; ++i) {
      status[g + i] = (uint8_t)st[i].status;
      Sponge sp = sha3_512_sponge();
      if (st[i].status == TX_OK) {
        const uint64_t head[3] = {st[i].n_in, st[i].n_out, (uint64_t)st[i].proof_len};
        sp.absorb((const uint8_t*)head, sizeof head);
        sp.absorb(st[i].commitments.data(), st[i].commitments.size());
        sp.absorb(st[i].sig_scalars.data(), st[i].sig_scalars.size());
        sp.absorb(st[i].sig_points.data() + 32, st[i].sig_points.size() - 32);     // ([0] is the basepoint's place, filled later)
      }
      sp.squeeze(digest + 64 * (g + i), 64);
    }
  }
}

// ---- the lockstep (AVX-512, eight transactions at a time) form of the payment VM's hashing against the one-at-a-time
// ---- form: for the CPU tests.  Per transaction out: status | txid (32) | then, for accepted ones, the statement's
// ---- commitments, signature scalars and points (after the challenge has been applied with the given aggregated keys)
// ---- folded into a SHA3-512 digest (64).  Returns 1 when AVX-512 is available (0: both modes ran one at a time).
extern "C" int zkhost_tx_prepare_group(const uint8_t* txs, const uint64_t* offs, size_t count, int lockstep, const uint8_t* agg_keys /*32 per tx*/,
                                       uint8_t* status, uint8_t* txid /*32 per tx*/, uint8_t* digest /*64 per tx*/) {
  using namespace zk::zkvm;
  uint8_t base[32];
  for (int i = 0; i < 32; ++i) base[i] = (uint8_t)(0xB0 + i);
  for (size_t g = 0; g < count; g += 8) {
    const size_t n = std::min<size_t>(8, count - g);
    const uint8_t* p[8]; size_t l[8];
    TxStatement st[8];
    for (size_t i = 0; i < n; ++i) { p[i] = txs + offs[g + i]; l[i] = (size_t)(offs[g + i + 1] - offs[g + i]); }
    tx_prepare_many(p, l, st, n, lockstep != 0);
    TxStatement* live[8]; const uint8_t* keys[8]; size_t nl = 0;
    for (size_t i = 0; i < n; ++i) if (st[i].status == TX_OK) { live[nl] = &st[i]; keys[nl] = agg_keys + 32 * (g + i); ++nl; }
    if (lockstep) tx_finish_signature_many(live, keys, base, nl);
    else for (size_t i = 0; i < nl; ++i) tx_finish_signature(*live[i], base, keys[i]);
    for (size_t i = 0; i < n; ++i) {
      status[g + i] = (uint8_t)st[i].status;
      std::memcpy(txid + 32 * (g + i), st[i].txid, 32);
      Sponge sp = sha3_512_sponge();
      if (st[i].status == TX_OK) {
        sp.absorb(st[i].commitments.data(), st[i].commitments.size());
        sp.absorb(st[i].sig_scalars.data(), st[i].sig_scalars.size());
        sp.absorb(st[i].sig_points.data(), st[i].sig_points.size());
      }
      sp.squeeze(digest + 64 * (g + i), 64);
    }
  }
  return x8_available() ? 1 : 0;
}

// ---- building transactions (zkvm_tx_build.hpp): what bench.py and the tests feed zkgpu_tx_verify_batch with -----------
#include "zkvm_tx_build.hpp"
#include <thread>

// one signed payment around an existing cloak proof -> its length (0: it does not fit `cap`, or the arities are outside 1..16)
extern "C" size_t zkhost_tx_wrap_payment(size_t n_in, size_t n_out, const uint8_t* com, const uint8_t* proof, size_t proof_len,
                                         const uint8_t seed[32], uint64_t mintime, uint64_t maxtime, uint8_t* out, size_t cap) {
  const std::vector<uint8_t> tx = zk::zkvm::tx_wrap_payment(n_in, n_out, com, proof, proof_len, seed, mintime, maxtime);
  if (tx.empty() || tx.size() > cap) return 0;
  std::memcpy(out, tx.data(), tx.size());
  return tx.size();
}

// `count` of them on `threads` threads: statement i = coms[i] (64 (n_in + n_out) bytes), proofs[i] (proof_len bytes), seeds[i]
// (32 bytes), mintime = mintime0 + i; the transactions back to back in `out`, offsets[count + 1].  -> 0, or -1 (one did not
// build or `cap` is too small)
extern "C" int zkhost_tx_wrap_many(size_t count, size_t n_in, size_t n_out, const uint8_t* coms, const uint8_t* proofs, size_t proof_len,
                                   const uint8_t* seeds, uint64_t mintime0, uint64_t maxtime, int threads, uint8_t* out, size_t cap,
                                   uint64_t* offsets) {
  std::vector<std::vector<uint8_t>> made(count);
  (void)zk::zkvm::base_table();                       // (built before the threads start)
  const int nt = std::max(1, std::min<int>(threads > 0 ? threads : zk::usable_cpus(), 64));
  std::vector<std::thread> th;
  const size_t wcom = 64 * (n_in + n_out);
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      for (size_t i = (size_t)t; i < count; i += (size_t)nt)
        made[i] = zk::zkvm::tx_wrap_payment(n_in, n_out, coms + i * wcom, proofs + i * proof_len, proof_len, seeds + 32 * i, mintime0 + i, maxtime);
    });
  for (auto& t : th) t.join();
  offsets[0] = 0;
  for (size_t i = 0; i < count; ++i) {
    if (made[i].empty()) return -1;
    offsets[i + 1] = offsets[i] + made[i].size();
  }
  if (offsets[count] > cap) return -1;
  for (size_t i = 0; i < count; ++i) std::memcpy(out + offsets[i], made[i].data(), made[i].size());
  return 0;
}

// ---- the scheduling of a transaction call on the CPU (tx_call.hpp) ----------------------------------------------------
// zkgpu_tx_verify_batch = TxCall + a device.  Here the device is a stand-in: every stage "runs" on a thread of its own that
// first sleeps a pseudo-random while (so that completions arrive in every order the real device could produce) and then
// computes the stage with this library's reference group arithmetic -- aggregated keys and signature equations for real,
// cloak proofs by a table the caller supplies (proof_ok[i]: this test is about the scheduling, and the reference R1CS
// verifier takes 30 ms per proof).  The sanitizer tier runs it under ThreadSanitizer and AddressSanitizer: flags, ring,
// stage cuts, hand-overs, the error paths (fail_at: the n-th device operation fails) -- no GPU involved.
#include "tx_call.hpp"
#include <atomic>
#include <random>

namespace {
using namespace zk::zkvm;

void ge_scalarmult_host(ge& out, const Scalar& s, const ge& p) {
  uint8_t b[32];
  s.to_bytes(b);
  ge_identity(out);
  for (int i = 255; i >= 0; --i) {
    ge_double(out, out);
    if ((b[i >> 3] >> (i & 7)) & 1) ge_add(out, out, p);
  }
}
bool decode_host(ge& p, const uint8_t enc[32]) {
  uint32_t w[8];
  std::memcpy(w, enc, 32);
  return ristretto_decode(p, w);
}

class HostTxDevice : public TxDevice {
 public:
  HostTxDevice(const uint8_t* txs, const uint64_t* offs, size_t batch, const uint8_t* proof_ok, uint32_t seed, int fail_at)
      : txs_(txs), offs_(offs), batch_(batch), proof_ok_(proof_ok), rng_(seed), fail_at_(fail_at) {
    ge B;
    B.X = fe_BASE_X(); B.Y = fe_BASE_Y(); B.Z = fe_one(); B.T = fe_BASE_T();
    encode_point(base_, B);
  }
  ~HostTxDevice() override {
    for (auto& s : keys_) if (s.th.joinable()) s.th.join();
    for (auto& s : sigs_) if (s.th.joinable()) s.th.join();
  }
  const uint8_t* basepoint() override { return base_; }
  int keys_enqueue(int slot, const uint8_t* sc, const uint8_t* pt, const uint64_t* off, size_t rows) override {
    if (failing()) return -3;
    Stage& s = keys_[slot];
    if (s.th.joinable()) { err_ = "key slot reused before it was collected"; return -1; }
    s.done = false;
    s.ok.assign((rows + 7) / 8 + 1, 0);
    s.values.assign(32 * rows, 0);
    const unsigned us = delay();
    s.th = std::thread([=, &s] {
      std::this_thread::sleep_for(std::chrono::microseconds(us));
      for (size_t r = 0; r < rows; ++r) {
        ge acc;
        ge_identity(acc);
        bool ok = true;
        for (uint64_t t = off[r]; t < off[r + 1]; ++t) {
          ge X, aX;
          Scalar a;
          if (!decode_host(X, pt + 32 * t) || !Scalar::from_canonical(sc + 32 * t, a)) { ok = false; break; }
          ge_scalarmult_host(aX, a, X);
          ge_add(acc, acc, aX);
        }
        if (ok) { s.ok[r / 8] |= (uint8_t)(1u << (r % 8)); encode_point(&s.values[32 * r], acc); }
      }
      s.done = true;
    });
    return 0;
  }
  bool keys_done(int slot) override { return keys_[slot].done; }
  int keys_collect(int slot, uint8_t* ok_bits, uint8_t* values) override {
    Stage& s = keys_[slot];
    if (!s.th.joinable()) { err_ = "nothing to collect in this key slot"; return -1; }
    s.th.join();
    if (failing()) return -3;
    std::memcpy(ok_bits, s.ok.data(), s.ok.size() - 1);
    std::memcpy(values, s.values.data(), s.values.size());
    return 0;
  }
  int proofs_stage(size_t ring_slot, size_t n, const TxProofSource* src, int, void** handle, std::string* err) override {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (ring_busy_[ring_slot]) { *err = "ring slot staged while its last chunk is still in flight"; return -1; }
      ring_busy_[ring_slot] = true;
      ++staged_total_;
    }
    Proofs* p = new Proofs();
    p->ring_slot = ring_slot;
    p->src.assign(src, src + n);
    *handle = p;
    return 0;
  }
  int proofs_start(size_t ring_slot, void* handle) override {
    Proofs* p = (Proofs*)handle;
    if (p->ring_slot != ring_slot) { err_ = "chunk started on another ring slot than it was staged in"; return -1; }
    if (failing()) return -3;
    p->bits.assign((p->src.size() + 7) / 8 + 1, 0);
    const unsigned us = delay();
    p->th = std::thread([this, p, us] {
      std::this_thread::sleep_for(std::chrono::microseconds(us));
      for (size_t q = 0; q < p->src.size(); ++q) {
        // which transaction of the call does this proof belong to?  (the proof lies inside its bytes)
        const uint64_t at = (uint64_t)(p->src[q].proof - txs_);
        size_t lo = 0, hi = batch_;
        while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (offs_[mid] <= at) lo = mid; else hi = mid; }
        if (proof_ok_[lo]) p->bits[q / 8] |= (uint8_t)(1u << (q % 8));
      }
      p->done = true;
    });
    return 0;
  }
  bool proofs_done(void* handle) override { return ((Proofs*)handle)->done; }
  int proofs_finish(void* handle, uint8_t* accept_bits) override {
    Proofs* p = (Proofs*)handle;
    if (p->th.joinable()) p->th.join();
    const bool fail = failing();
    if (!fail) std::memcpy(accept_bits, p->bits.data(), p->bits.size() - 1);
    proofs_release(handle);
    return fail ? -3 : 0;
  }
  void proofs_release(void* handle) override {
    Proofs* p = (Proofs*)handle;
    if (p->th.joinable()) p->th.join();
    { std::lock_guard<std::mutex> lk(mu_); ring_busy_[p->ring_slot] = false; ++released_total_; }
    delete p;
  }
  int sigs_enqueue(int slot, size_t rows, const uint8_t* dsc, const uint8_t* dpt, const uint64_t* doff, const uint8_t* bsc) override {
    if (failing()) return -3;
    Stage& s = sigs_[slot];
    if (s.th.joinable()) { err_ = "signature slot reused before it was collected"; return -1; }
    s.done = false;
    s.ok.assign((rows + 7) / 8 + 1, 0);
    const unsigned us = delay();
    s.th = std::thread([=, &s] {
      std::this_thread::sleep_for(std::chrono::microseconds(us));
      for (size_t r = 0; r < rows; ++r) {
        Scalar sb;
        if (!Scalar::from_canonical(bsc + 32 * r, sb)) continue;
        ge acc;
        base_mul(acc, sb);
        bool ok = true;
        for (uint64_t t = doff[r]; t < doff[r + 1]; ++t) {
          ge X, aX;
          Scalar a;
          if (!decode_host(X, dpt + 32 * t) || !Scalar::from_canonical(dsc + 32 * t, a)) { ok = false; break; }
          ge_scalarmult_host(aX, a, X);
          ge_add(acc, acc, aX);
        }
        if (ok && ge_is_identity(acc)) s.ok[r / 8] |= (uint8_t)(1u << (r % 8));
      }
      s.done = true;
    });
    return 0;
  }
  bool sigs_done(int slot) override { return sigs_[slot].done; }
  int sigs_collect(int slot, uint8_t* bits) override {
    Stage& s = sigs_[slot];
    if (!s.th.joinable()) { err_ = "nothing to collect in this signature slot"; return -1; }
    s.th.join();
    if (failing()) return -3;
    std::memcpy(bits, s.ok.data(), s.ok.size() - 1);
    return 0;
  }
  std::string last_error() override { return err_.empty() ? "injected device fault" : err_; }
  size_t leaked() { std::lock_guard<std::mutex> lk(mu_); return staged_total_ - released_total_; }

 private:
  struct Stage { std::thread th; std::atomic<bool> done{false}; std::vector<uint8_t> ok, values; };
  struct Proofs { size_t ring_slot = 0; std::vector<TxProofSource> src; std::vector<uint8_t> bits; std::thread th; std::atomic<bool> done{false}; };
  unsigned delay() { std::lock_guard<std::mutex> lk(mu_); return (unsigned)(rng_() % 400); }
  bool failing() { return fail_at_ >= 0 && ops_.fetch_add(1) == fail_at_; }
  const uint8_t* txs_;
  const uint64_t* offs_;
  size_t batch_;
  const uint8_t* proof_ok_;
  std::mutex mu_;
  std::mt19937 rng_;
  const int fail_at_;
  std::atomic<int> ops_{0};
  Stage keys_[2], sigs_[2];
  bool ring_busy_[TxCall::RING] = {false};
  size_t staged_total_ = 0, released_total_ = 0;
  uint8_t base_[32];
  std::string err_;
};
}  // namespace

// -> the call's return code; *n_chunks, *n_sig_stages: what was planned; *leaked: staged chunks never released (must be 0)
extern "C" int zkhost_txcall_selftest(size_t batch, const uint8_t* txs, const uint64_t* offs, const uint8_t* proof_ok, int host_threads,
                                      size_t chunk, uint32_t delay_seed, int fail_at, uint8_t* accept_bitmap, uint8_t* status,
                                      size_t* n_chunks, size_t* n_sig_stages, size_t* leaked) {
  std::memset(accept_bitmap, 0, (batch + 7) / 8);
  std::memset(status, TX_INVALID, batch);
  std::vector<TxStatement> store;
  HostTxDevice dev(txs, offs, batch, proof_ok, delay_seed, fail_at);
  int rc;
  {
    TxCall call(dev, store, (size_t)1 << 17, batch, txs, offs, host_threads, chunk, accept_bitmap, status);
    *n_chunks = call.n_chunks();
    *n_sig_stages = call.n_sig_stages_planned();
    rc = call.run();
  }
  *leaked = dev.leaked();
  if (rc != 0) std::memset(accept_bitmap, 0, (batch + 7) / 8);
  return rc;
}

// TWO calls driven by ONE thread through start / step / done / finish, one stage slot each -- how the engine of
// zkgpu_tx_verify_submit keeps two rounds in flight: the transactions [0, split) and [split, batch) as two calls whose steps
// alternate; the caller naps on a condition variable the calls' staging threads signal (on_news).  -> 0, or the first error
extern "C" int zkhost_txcall_pair_selftest(size_t batch, size_t split, const uint8_t* txs, const uint64_t* offs, const uint8_t* proof_ok,
                                           int host_threads, size_t chunk, uint32_t delay_seed, uint8_t* accept_bitmap, uint8_t* status,
                                           size_t* leaked) {
  std::memset(accept_bitmap, 0, (batch + 7) / 8);
  std::memset(status, TX_INVALID, batch);
  if (split == 0 || split >= batch || split % 8) return -1;
  std::vector<TxStatement> store[2];
  std::vector<uint8_t> bits[2] = {std::vector<uint8_t>((split + 7) / 8 + 1, 0), std::vector<uint8_t>((batch - split + 7) / 8 + 1, 0)};
  std::mutex news_mu;
  std::condition_variable news_cv;
  bool news = false;
  int rc_all = 0;
  size_t leaked_all = 0;
  {
    HostTxDevice dev0(txs, offs, batch, proof_ok, delay_seed, -1), dev1(txs, offs, batch, proof_ok, delay_seed + 1, -1);
    std::vector<TxCall::Piece> p0{{txs, offs, split}}, p1{{txs, offs + split, batch - split}};
    // (the second call's offsets are still relative to `txs`: Piece = base pointer + its own offsets)
    TxCall c0(dev0, store[0], (size_t)1 << 17, p0, host_threads, chunk, bits[0].data(), status, 1);
    TxCall c1(dev1, store[1], (size_t)1 << 17, p1, host_threads, chunk, bits[1].data(), status + split, 1);
    TxCall* calls[2] = {&c0, &c1};
    for (TxCall* c : calls) {
      c->set_on_news([&] { { std::lock_guard<std::mutex> nl(news_mu); news = true; } news_cv.notify_one(); });
      const int rc = c->start();
      if (rc != 0 && rc_all == 0) rc_all = rc;
    }
    while (!c0.done() || !c1.done()) {
      bool progress = false;
      for (TxCall* c : calls) if (!c->done()) progress |= c->step();
      if (!progress) {
        std::unique_lock<std::mutex> nl(news_mu);
        if (!news) news_cv.wait_until(nl, std::chrono::system_clock::now() + std::chrono::microseconds(50));
        news = false;
      }
    }
    // (the engine's rule: finish() only once the device has settled, so that it never sleeps inside one call)
    for (int spins = 0; !(c0.settled() && c1.settled()) && spins < 2000000; ++spins) std::this_thread::sleep_for(std::chrono::microseconds(20));
    for (TxCall* c : calls) { const int rc = c->finish(); if (rc != 0 && rc_all == 0) rc_all = rc; }
    leaked_all = dev0.leaked() + dev1.leaked();
  }
  *leaked = leaked_all;
  if (rc_all != 0) return rc_all;
  for (size_t i = 0; i < split; ++i) if ((bits[0][i / 8] >> (i % 8)) & 1) accept_bitmap[i / 8] |= (uint8_t)(1u << (i % 8));
  for (size_t i = split; i < batch; ++i) { const size_t q = i - split; if ((bits[1][q / 8] >> (q % 8)) & 1) accept_bitmap[i / 8] |= (uint8_t)(1u << (i % 8)); }
  return 0;
}
