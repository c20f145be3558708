/* cloak.c -- the "cloak" constraint gadget (merge / split / shuffle / range proof)
 * and a transaction-shaped statement built on it.
 *
 * TEST INFRASTRUCTURE (see oracle.h).  Restates the `spacesuit` crate's `cloak`
 * (SURVEY.md sec 2 "spacesuit::cloak gadgets", sec 8 f-2/f-3; source NOT mounted)
 * from the published Cloak specification as recalled: a k-value shuffle is a
 * scalar shuffle of q + w f, a scalar shuffle is the polynomial identity
 * prod (x_i - z) = prod (y_i - z), a mix either passes (A,B) through or merges
 * them into ((0,0), (A.q + B.q, A.f)), a range proof is a 64-bit decomposition.
 * PARITY UNPINNED (layout and challenge labels cannot be checked here); its role
 * is to give the verification path real proofs of the reference's *shape*:
 * 2-in/2-out -> 150 multipliers, padded n = 256, k = 8, m = 8 commitments.
 */
#include "r1cs.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void lc_var(r1cs_lc *l, r1cs_var v) { lc_init(l); lc_push_u64(l, v, 1); }
static void lc_add_var(r1cs_lc *l, r1cs_var v, const sc *c) { lc_push(l, v, c); }
static void lc_sub_var(r1cs_lc *l, r1cs_var v, const sc *c) { sc n; sc_neg(&n, c); lc_push(l, v, &n); }

/* ---- scalar shuffle: prod (x_i - z) == prod (y_i - z) ---------------------------- */
typedef struct { size_t k; r1cs_var *x, *y; } shuffle_ud;

static r1cs_var product_minus_z(r1cs_cs *cs, const r1cs_var *x, size_t k, const sc *z) {
  sc one; sc_from_u64(&one, 1);
  r1cs_lc a, b;
  r1cs_var out[3];
  lc_var(&a, x[k - 1]); lc_sub_var(&a, var_one(), z);
  lc_var(&b, x[k - 2]); lc_sub_var(&b, var_one(), z);
  r1cs_multiply(cs, &a, &b, out);
  for (size_t i = k - 2; i-- > 0;) {
    lc_var(&a, out[2]);
    lc_var(&b, x[i]); lc_sub_var(&b, var_one(), z);
    r1cs_multiply(cs, &a, &b, out);
  }
  return out[2];
}

static int scalar_shuffle_cb(r1cs_cs *cs, void *p) {
  shuffle_ud *ud = p;
  sc z, one;
  sc_from_u64(&one, 1);
  r1cs_challenge_scalar(cs, "shuffle challenge", &z);
  r1cs_var px = product_minus_z(cs, ud->x, ud->k, &z);
  r1cs_var py = product_minus_z(cs, ud->y, ud->k, &z);
  r1cs_lc l;
  lc_var(&l, px); lc_sub_var(&l, py, &one);
  r1cs_constrain(cs, &l);
  free(ud->x); free(ud->y); free(ud);
  return 0;
}

static int scalar_shuffle(r1cs_cs *cs, const r1cs_var *x, const r1cs_var *y, size_t k) {
  sc one; sc_from_u64(&one, 1);
  if (k == 0) return 0;
  if (k == 1) {
    r1cs_lc l; lc_var(&l, y[0]); lc_sub_var(&l, x[0], &one);
    r1cs_constrain(cs, &l);
    return 0;
  }
  shuffle_ud *ud = malloc(sizeof *ud);
  ud->k = k;
  ud->x = malloc(sizeof(r1cs_var) * k); ud->y = malloc(sizeof(r1cs_var) * k);
  memcpy(ud->x, x, sizeof(r1cs_var) * k); memcpy(ud->y, y, sizeof(r1cs_var) * k);
  return r1cs_specify_randomized_constraints(cs, scalar_shuffle_cb, ud);
}

/* ---- value shuffle: scalar shuffle of q + w f -------------------------------------- */
typedef struct { size_t k; cloak_value *x, *y; } vshuffle_ud;

static int value_shuffle_cb(r1cs_cs *cs, void *p) {
  vshuffle_ud *ud = p;
  sc w;
  r1cs_challenge_scalar(cs, "k-value shuffle challenge", &w);
  r1cs_var *xs = malloc(sizeof(r1cs_var) * ud->k), *ys = malloc(sizeof(r1cs_var) * ud->k);
  for (size_t i = 0; i < ud->k; ++i) {
    r1cs_lc a, b;
    r1cs_var out[3];
    lc_var(&a, ud->x[i].q); lc_add_var(&a, ud->x[i].f, &w);
    lc_var(&b, ud->y[i].q); lc_add_var(&b, ud->y[i].f, &w);
    r1cs_multiply(cs, &a, &b, out);
    xs[i] = out[0]; ys[i] = out[1];
  }
  int rc = scalar_shuffle(cs, xs, ys, ud->k);
  free(xs); free(ys); free(ud->x); free(ud->y); free(ud);
  return rc;
}

static int value_shuffle(r1cs_cs *cs, const cloak_value *x, const cloak_value *y, size_t k) {
  sc one; sc_from_u64(&one, 1);
  if (k == 0) return 0;
  if (k == 1) {
    r1cs_lc l;
    lc_var(&l, x[0].q); lc_sub_var(&l, y[0].q, &one); r1cs_constrain(cs, &l);
    lc_var(&l, x[0].f); lc_sub_var(&l, y[0].f, &one); r1cs_constrain(cs, &l);
    return 0;
  }
  vshuffle_ud *ud = malloc(sizeof *ud);
  ud->k = k;
  ud->x = malloc(sizeof(cloak_value) * k); ud->y = malloc(sizeof(cloak_value) * k);
  memcpy(ud->x, x, sizeof(cloak_value) * k); memcpy(ud->y, y, sizeof(cloak_value) * k);
  return r1cs_specify_randomized_constraints(cs, value_shuffle_cb, ud);
}

static cloak_value allocate_value(r1cs_cs *cs, uint64_t q, const sc *f, int has) {
  cloak_value v;
  r1cs_var out[3];
  sc qs; sc_from_u64(&qs, q);
  sc fz; sc_from_u64(&fz, 0);
  r1cs_allocate_multiplier(cs, &qs, has ? f : &fz, out);
  v.q = out[0]; v.f = out[1]; v.q_val = q; v.f_val = has ? *f : fz; v.has_assignment = has;
  return v;
}

static int padded_shuffle(r1cs_cs *cs, const cloak_value *x, size_t kx, const cloak_value *y, size_t ky) {
  size_t k = kx > ky ? kx : ky;
  cloak_value *xp = malloc(sizeof(cloak_value) * (k + 1)), *yp = malloc(sizeof(cloak_value) * (k + 1));
  sc zero; sc_from_u64(&zero, 0);
  int has = r1cs_is_prover(cs);
  for (size_t i = 0; i < k; ++i) xp[i] = i < kx ? x[i] : allocate_value(cs, 0, &zero, has);
  for (size_t i = 0; i < k; ++i) yp[i] = i < ky ? y[i] : allocate_value(cs, 0, &zero, has);
  int rc = value_shuffle(cs, xp, yp, k);
  free(xp); free(yp);
  return rc;
}

/* ---- mix: (A, B) -> (C, D), unchanged or merged --------------------------------------- */
typedef struct { cloak_value A, B, C, D; } mix_ud;

static int mix_cb(r1cs_cs *cs, void *p) {
  mix_ud *m = p;
  sc w, w2, w3, w4, one;
  sc_from_u64(&one, 1);
  r1cs_challenge_scalar(cs, "mix challenge", &w);
  sc_mul(&w2, &w, &w); sc_mul(&w3, &w2, &w); sc_mul(&w4, &w3, &w);
  r1cs_lc l, r;
  /* (A.q - C.q) + (A.f - C.f) w + (B.q - D.q) w^2 + (B.f - D.f) w^3 */
  lc_init(&l);
  lc_add_var(&l, m->A.q, &one); lc_sub_var(&l, m->C.q, &one);
  lc_add_var(&l, m->A.f, &w); lc_sub_var(&l, m->C.f, &w);
  lc_add_var(&l, m->B.q, &w2); lc_sub_var(&l, m->D.q, &w2);
  lc_add_var(&l, m->B.f, &w3); lc_sub_var(&l, m->D.f, &w3);
  /* C.q + (A.f - B.f) w^4 + (D.q - A.q - B.q) w^2 + (D.f - A.f) w^3 */
  lc_init(&r);
  lc_add_var(&r, m->C.q, &one);
  lc_add_var(&r, m->A.f, &w4); lc_sub_var(&r, m->B.f, &w4);
  lc_add_var(&r, m->D.q, &w2); lc_sub_var(&r, m->A.q, &w2); lc_sub_var(&r, m->B.q, &w2);
  lc_add_var(&r, m->D.f, &w3); lc_sub_var(&r, m->A.f, &w3);
  r1cs_var out[3];
  r1cs_multiply(cs, &l, &r, out);
  r1cs_lc o; lc_var(&o, out[2]);
  r1cs_constrain(cs, &o);
  free(m);
  return 0;
}

static int sc_cmp(const sc *a, const sc *b) {
  for (int i = 3; i >= 0; --i) { if (a->v[i] != b->v[i]) return a->v[i] < b->v[i] ? -1 : 1; }
  return 0;
}

/* k_mix: allocate `grouped` (the values ordered so equal flavors are adjacent) and
 * `merged` (one value per flavor run carrying the run's total, zeros elsewhere) and chain mixes. */
static int k_mix(r1cs_cs *cs, const cloak_value *vals, size_t k, cloak_value *grouped, cloak_value *merged) {
  int has = r1cs_is_prover(cs);
  sc zero; sc_from_u64(&zero, 0);
  if (k == 0) return 0;
  if (k == 1) { grouped[0] = vals[0]; merged[0] = vals[0]; return 0; }
  /* witness: order by flavor (insertion sort, stable) */
  uint64_t *gq = malloc(sizeof(uint64_t) * k), *mq = malloc(sizeof(uint64_t) * k), *dq = malloc(sizeof(uint64_t) * k);
  sc *gf = malloc(sizeof(sc) * k), *mf = malloc(sizeof(sc) * k), *df = malloc(sizeof(sc) * k);
  for (size_t i = 0; i < k; ++i) { gq[i] = has ? vals[i].q_val : 0; gf[i] = has ? vals[i].f_val : zero; }
  if (has) {
    for (size_t i = 1; i < k; ++i) {
      uint64_t q = gq[i]; sc f = gf[i]; size_t j = i;
      while (j > 0 && sc_cmp(&gf[j - 1], &f) > 0) { gq[j] = gq[j - 1]; gf[j] = gf[j - 1]; --j; }
      gq[j] = q; gf[j] = f;
    }
    uint64_t aq = gq[0]; sc af = gf[0];
    for (size_t i = 1; i < k; ++i) {
      if (sc_cmp(&af, &gf[i]) == 0) { mq[i - 1] = 0; mf[i - 1] = zero; aq += gq[i]; }
      else { mq[i - 1] = aq; mf[i - 1] = af; aq = gq[i]; af = gf[i]; }
      dq[i - 1] = aq; df[i - 1] = af;   /* running D after step i */
    }
    mq[k - 1] = aq; mf[k - 1] = af;
  } else {
    for (size_t i = 0; i < k; ++i) { mq[i] = dq[i] = 0; mf[i] = df[i] = zero; }
  }
  for (size_t i = 0; i < k; ++i) grouped[i] = allocate_value(cs, gq[i], &gf[i], has);
  cloak_value *mid = malloc(sizeof(cloak_value) * k);
  for (size_t i = 0; i + 2 < k; ++i) mid[i] = allocate_value(cs, dq[i], &df[i], has);
  for (size_t i = 0; i < k; ++i) merged[i] = allocate_value(cs, mq[i], &mf[i], has);
  for (size_t i = 0; i + 1 < k; ++i) {
    mix_ud *m = malloc(sizeof *m);
    m->A = i == 0 ? grouped[0] : mid[i - 1];
    m->B = grouped[i + 1];
    m->C = merged[i];
    m->D = (i + 2 == k) ? merged[k - 1] : mid[i];
    int rc = r1cs_specify_randomized_constraints(cs, mix_cb, m);
    if (rc) return rc;
  }
  free(gq); free(mq); free(dq); free(gf); free(mf); free(df); free(mid);
  return 0;
}

/* ---- 64-bit range proof on a quantity -------------------------------------------------- */
static void range_proof(r1cs_cs *cs, r1cs_var v, uint64_t q, int has, int nbits) {
  sc one, exp2, neg;
  sc_from_u64(&one, 1);
  exp2 = one;
  r1cs_lc acc;
  lc_var(&acc, v);
  for (int i = 0; i < nbits; ++i) {
    uint64_t bit = has ? (q >> i) & 1 : 0;
    sc a, b;
    sc_from_u64(&a, 1 - bit); sc_from_u64(&b, bit);
    r1cs_var out[3];
    r1cs_allocate_multiplier(cs, &a, &b, out);
    r1cs_lc l;
    lc_var(&l, out[2]); r1cs_constrain(cs, &l);                       /* a * b = 0 */
    lc_var(&l, out[0]); lc_add_var(&l, out[1], &one); lc_sub_var(&l, var_one(), &one);
    r1cs_constrain(cs, &l);                                           /* a + b - 1 = 0 */
    sc_neg(&neg, &exp2);
    lc_push(&acc, out[1], &neg);                                      /* v - sum b_i 2^i */
    sc_add(&exp2, &exp2, &exp2);
  }
  r1cs_constrain(cs, &acc);
}

int cloak_gadget(r1cs_cs *cs, const cloak_value *in, size_t n_in, const cloak_value *out, size_t n_out) {
  cloak_value *merge_in = malloc(sizeof(cloak_value) * (n_in + 1)), *merge_out = malloc(sizeof(cloak_value) * (n_in + 1));
  cloak_value *split_out = malloc(sizeof(cloak_value) * (n_out + 1)), *split_in = malloc(sizeof(cloak_value) * (n_out + 1));
  int rc = k_mix(cs, in, n_in, merge_in, merge_out);
  if (!rc) rc = k_mix(cs, out, n_out, split_out, split_in);
  if (!rc) rc = value_shuffle(cs, in, merge_in, n_in);
  if (!rc) rc = padded_shuffle(cs, merge_out, n_in, split_in, n_out);
  if (!rc) rc = value_shuffle(cs, split_out, out, n_out);
  if (!rc)
    for (size_t i = 0; i < n_out; ++i) range_proof(cs, out[i].q, out[i].q_val, out[i].has_assignment, 64);
  free(merge_in); free(merge_out); free(split_out); free(split_in);
  return rc;
}

/* ---- transaction-shaped statement ---------------------------------------------------------- */
static const char TX_LABEL[] = "ZkVM.r1cs";

static void derive(const uint8_t seed[32], const char *tag, uint64_t i, uint8_t *out, size_t n) {
  shake256_ctx c;
  uint8_t ib[8];
  for (int k = 0; k < 8; ++k) ib[k] = (uint8_t)(i >> (8 * k));
  shake256_init(&c);
  shake256_absorb(&c, seed, 32);
  shake256_absorb(&c, (const uint8_t *)tag, strlen(tag));
  shake256_absorb(&c, ib, 8);
  shake256_squeeze(&c, out, n);
}

int zko_cloak_prove(const uint64_t *q, const uint8_t *flavors, size_t n_in, size_t n_out, const uint8_t seed[32],
                    uint8_t *commitments, uint8_t *proof, size_t proof_cap, size_t *proof_len,
                    size_t *n_multipliers) {
  size_t nv = n_in + n_out;
  r1cs_cs *cs = r1cs_prover_new((const uint8_t *)TX_LABEL, strlen(TX_LABEL));
  cloak_value *vals = malloc(sizeof(cloak_value) * (nv + 1));
  for (size_t i = 0; i < nv; ++i) {
    sc qs, fs, bl;
    uint8_t wide[64];
    sc_from_u64(&qs, q[i]);
    sc_from_bytes_mod_order(&fs, flavors + 32 * i);
    derive(seed, "q_blinding", i, wide, 64); sc_from_bytes_wide(&bl, wide);
    vals[i].q = r1cs_prover_commit(cs, &qs, &bl, commitments + 64 * i);
    derive(seed, "f_blinding", i, wide, 64); sc_from_bytes_wide(&bl, wide);
    vals[i].f = r1cs_prover_commit(cs, &fs, &bl, commitments + 64 * i + 32);
    vals[i].q_val = q[i]; vals[i].f_val = fs; vals[i].has_assignment = 1;
  }
  int rc = cloak_gadget(cs, vals, n_in, vals + n_in, n_out);
  uint8_t rng_seed[32];
  derive(seed, "rng", 0, rng_seed, 32);
  if (!rc) rc = r1cs_prove(cs, rng_seed, proof, proof_cap, proof_len);
  if (n_multipliers) *n_multipliers = r1cs_num_multipliers(cs);
  free(vals);
  r1cs_free(cs);
  return rc;
}

static r1cs_cs *cloak_verifier(const uint8_t *commitments, size_t n_in, size_t n_out) {
  size_t nv = n_in + n_out;
  r1cs_cs *cs = r1cs_verifier_new((const uint8_t *)TX_LABEL, strlen(TX_LABEL));
  cloak_value *vals = malloc(sizeof(cloak_value) * (nv + 1));
  for (size_t i = 0; i < nv; ++i) {
    vals[i].q = r1cs_verifier_commit(cs, commitments + 64 * i);
    vals[i].f = r1cs_verifier_commit(cs, commitments + 64 * i + 32);
    vals[i].q_val = 0; sc_from_u64(&vals[i].f_val, 0); vals[i].has_assignment = 0;
  }
  int rc = cloak_gadget(cs, vals, n_in, vals + n_in, n_out);
  free(vals);
  if (rc) { r1cs_free(cs); return NULL; }
  return cs;
}

int zko_cloak_verify(const uint8_t *commitments, size_t n_in, size_t n_out, const uint8_t *proof, size_t proof_len,
                     const uint8_t r_bytes[64]) {
  r1cs_cs *cs = cloak_verifier(commitments, n_in, n_out);
  if (!cs) return 0;
  int ok = r1cs_verify(cs, proof, proof_len, r_bytes);
  r1cs_free(cs);
  return ok;
}

int zko_cloak_verify_prepare(const uint8_t *commitments, size_t n_in, size_t n_out, const uint8_t *proof,
                             size_t proof_len, const uint8_t r_bytes[64], r1cs_msm *out) {
  r1cs_cs *cs = cloak_verifier(commitments, n_in, n_out);
  if (!cs) return -1;
  int rc = r1cs_verify_prepare(cs, proof, proof_len, r_bytes, out);
  r1cs_free(cs);
  return rc;
}

int zko_cloak_verify_challenges(const uint8_t *commitments, size_t n_in, size_t n_out, const uint8_t *proof,
                                size_t proof_len, const uint8_t r_bytes[64], uint8_t *challenges, size_t cap,
                                size_t *n_challenges) {
  r1cs_cs *cs = cloak_verifier(commitments, n_in, n_out);
  if (!cs) return -1;
  r1cs_msm m;
  int rc = r1cs_verify_prepare(cs, proof, proof_len, r_bytes, &m);
  if (rc == 0) { *n_challenges = r1cs_challenge_log(cs, challenges, cap); r1cs_msm_free(&m); }
  r1cs_free(cs);
  return rc;
}

/* count proofs of one shape with seeded witnesses: tx i moves q0, q1 of flavor(s) chosen by i:
 * even i: one flavor (merge + split exercised), odd i: two flavors (pass-through). */
int zko_cloak_prove_batch(size_t count, size_t n_in, size_t n_out, const uint8_t seed[32], uint8_t *commitments,
                          uint8_t *proofs, size_t proof_stride, size_t *proof_len, int threads) {
  int failed = 0;
  size_t nv = n_in + n_out;
  size_t plen = 0;
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 1 ? threads : 1)
#endif
  for (long long i = 0; i < (long long)count; ++i) {
    uint8_t s[32], fl[2][32], raw[8];
    derive(seed, "tx", (uint64_t)i, s, 32);
    derive(s, "flavor", 0, fl[0], 32); fl[0][31] &= 0x0f;
    derive(s, "flavor", 1, fl[1], 32); fl[1][31] &= 0x0f;
    uint64_t *q = malloc(sizeof(uint64_t) * nv);
    uint8_t *f = malloc(32 * nv);
    /* balanced witness: inputs random 40-bit amounts; outputs redistribute per flavor */
    uint64_t tot[2] = {0, 0};
    const int two = (i & 1) && n_in >= 2 && n_out >= 2;
    for (size_t j = 0; j < n_in; ++j) {
      derive(s, "amount", j, raw, 8);
      uint64_t a = 0;
      for (int k = 0; k < 5; ++k) a |= (uint64_t)raw[k] << (8 * k);
      int fi = two ? (int)(j & 1) : 0;
      q[j] = a; memcpy(f + 32 * j, fl[fi], 32); tot[fi] += a;
    }
    for (size_t j = 0; j < n_out; ++j) {
      int fi = two ? (int)(j & 1) : 0;
      /* last output of a flavor takes the remainder */
      int last = 1;
      for (size_t jj = j + 1; jj < n_out; ++jj) if ((two ? (int)(jj & 1) : 0) == fi) last = 0;
      uint64_t a = last ? tot[fi] : tot[fi] / 3;
      q[n_in + j] = a; tot[fi] -= a; memcpy(f + 32 * (n_in + j), fl[fi], 32);
    }
    size_t len = 0;
    int rc = zko_cloak_prove(q, f, n_in, n_out, s, commitments + 64 * nv * (size_t)i, proofs + proof_stride * (size_t)i,
                             proof_stride, &len, NULL);
    if (rc || tot[0] || tot[1]) {
#pragma omp atomic
      failed++;
    }
    if (i == 0) plen = len;
    free(q); free(f);
  }
  if (proof_len) *proof_len = plen;
  return failed ? -1 : 0;
}

/* verify `count` proofs of one shape (OpenMP over proofs): accept[i] = 1 / 0 */
int zko_cloak_verify_batch(size_t count, size_t n_in, size_t n_out, const uint8_t *commitments, const uint8_t *proofs,
                           size_t proof_stride, size_t proof_len, const uint8_t *r_bytes, uint8_t *accept,
                           int threads) {
  size_t w = 64 * (n_in + n_out);
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 1 ? threads : 1)
#endif
  for (long long i = 0; i < (long long)count; ++i)
    accept[i] = (uint8_t)zko_cloak_verify(commitments + w * (size_t)i, n_in, n_out, proofs + proof_stride * (size_t)i,
                                           proof_len, r_bytes + 64 * (size_t)i);
  return 0;
}
