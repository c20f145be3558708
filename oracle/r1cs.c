/* r1cs.c -- Bulletproofs R1CS constraint system, prover and verifier (CPU).
 *
 * TEST INFRASTRUCTURE (see oracle.h).  Restates the `bulletproofs` crate's
 * `r1cs::{Prover, Verifier, R1CSProof, LinearCombination}` and
 * `inner_product_proof::InnerProductProof` with the `yoloproofs` two-phase
 * extension (the Rust source is NOT mounted
 * under /root/reference).  Written from Bunz et al. 2018 sec 3 and sec 5 and the
 * crate's published "R1CS proof protocol" notes.
 *
 * PARITY UNPINNED for this layer: transcript labels, message order and the wire
 * format follow upstream as recalled and nothing here can
 * check them.  What the tests do pin is self-consistency (prove -> verify,
 * every mutation rejected) and that the verification equation this file builds
 * is the one libzkgpu evaluates (same accept bits on the same proof bytes).
 */
#include "r1cs.h"
#include <stdlib.h>
#include <string.h>

/* ---- generator cache ------------------------------------------------------ */
static ge *g_cacheG = NULL, *g_cacheH = NULL;
static size_t g_cache_n = 0;

static void gens_ensure(size_t n) {
#pragma omp critical(zko_gens)
  {
    if (n > g_cache_n) {
      size_t cap = 64;
      while (cap < n) cap *= 2;
      ge *G = malloc(sizeof(ge) * cap), *H = malloc(sizeof(ge) * cap);
      bulletproof_gens_chain(G, cap, 'G', 0);
      bulletproof_gens_chain(H, cap, 'H', 0);
      /* older pointers may still be in use by other threads: leak them (test infra) */
      g_cacheG = G; g_cacheH = H; g_cache_n = cap;
    }
  }
}

/* ---- linear combinations -------------------------------------------------- */
void lc_init(r1cs_lc *l) { l->t = NULL; l->n = l->cap = 0; }
void lc_free(r1cs_lc *l) { free(l->t); lc_init(l); }
void lc_push(r1cs_lc *l, r1cs_var v, const sc *c) {
  if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 8; l->t = realloc(l->t, sizeof(r1cs_term) * l->cap); }
  l->t[l->n].v = v; l->t[l->n].c = *c; l->n++;
}
void lc_push_u64(r1cs_lc *l, r1cs_var v, uint64_t c) { sc s; sc_from_u64(&s, c); lc_push(l, v, &s); }
void lc_push_neg_u64(r1cs_lc *l, r1cs_var v, uint64_t c) { sc s; sc_from_u64(&s, c); sc_neg(&s, &s); lc_push(l, v, &s); }
void lc_add_scaled(r1cs_lc *dst, const r1cs_lc *src, const sc *scale) {
  for (size_t i = 0; i < src->n; ++i) { sc c; sc_mul(&c, &src->t[i].c, scale); lc_push(dst, src->t[i].v, &c); }
}
void lc_clone(r1cs_lc *dst, const r1cs_lc *src) {
  lc_init(dst);
  for (size_t i = 0; i < src->n; ++i) lc_push(dst, src->t[i].v, &src->t[i].c);
}
r1cs_var var_one(void) { r1cs_var v = {VAR_ONE, 0}; return v; }

/* ---- constraint system ----------------------------------------------------- */
struct r1cs_cs {
  int is_prover, phase;
  merlin_transcript tr;
  ge B, B_blinding;
  r1cs_lc *cons; size_t n_cons, cap_cons;
  size_t num_vars; long pending;
  sc *aL, *aR, *aO; size_t cap_vars;
  sc *v, *v_blinding; uint8_t (*V)[32]; size_t m, cap_m;
  struct { r1cs_rand_fn fn; void *ud; } *deferred; size_t n_def, cap_def;
  /* every Fiat-Shamir challenge drawn through this system, in transcript order (tests compare the
   * device-side transcript replay with these byte for byte) */
  sc *chal_log; size_t n_chal, cap_chal;
};

static void draw_challenge(r1cs_cs *cs, const char *label, sc *out) {
  merlin_challenge_scalar(&cs->tr, label, out);
  if (cs->n_chal == cs->cap_chal) {
    cs->cap_chal = cs->cap_chal ? 2 * cs->cap_chal : 32;
    cs->chal_log = realloc(cs->chal_log, sizeof(sc) * cs->cap_chal);
  }
  cs->chal_log[cs->n_chal++] = *out;
}

size_t r1cs_challenge_log(const r1cs_cs *cs, uint8_t *out, size_t cap) {
  size_t n = cs->n_chal < cap ? cs->n_chal : cap;
  for (size_t i = 0; i < n; ++i) sc_to_bytes(out + 32 * i, &cs->chal_log[i]);
  return cs->n_chal;
}

static r1cs_cs *cs_new(int is_prover, const uint8_t *label, size_t len) {
  r1cs_cs *cs = calloc(1, sizeof *cs);
  cs->is_prover = is_prover; cs->phase = 1; cs->pending = -1;
  merlin_init(&cs->tr, label, len);
  merlin_append_message(&cs->tr, "dom-sep", (const uint8_t *)"r1cs v1", 7);
  pedersen_gens(&cs->B, &cs->B_blinding);
  return cs;
}
r1cs_cs *r1cs_prover_new(const uint8_t *label, size_t len) { return cs_new(1, label, len); }
r1cs_cs *r1cs_verifier_new(const uint8_t *label, size_t len) { return cs_new(0, label, len); }

void r1cs_free(r1cs_cs *cs) {
  if (!cs) return;
  for (size_t i = 0; i < cs->n_cons; ++i) lc_free(&cs->cons[i]);
  free(cs->cons); free(cs->aL); free(cs->aR); free(cs->aO); free(cs->v); free(cs->v_blinding); free(cs->V);
  free(cs->deferred); free(cs->chal_log); free(cs);
}

static void grow_m(r1cs_cs *cs) {
  if (cs->m == cs->cap_m) {
    cs->cap_m = cs->cap_m ? 2 * cs->cap_m : 16;
    cs->v = realloc(cs->v, sizeof(sc) * cs->cap_m);
    cs->v_blinding = realloc(cs->v_blinding, sizeof(sc) * cs->cap_m);
    cs->V = realloc(cs->V, 32 * cs->cap_m);
  }
}

r1cs_var r1cs_prover_commit(r1cs_cs *cs, const sc *v, const sc *v_blinding, uint8_t out[32]) {
  grow_m(cs);
  ge a, b, c;
  ge_scalarmult(&a, v, &cs->B);
  ge_scalarmult(&b, v_blinding, &cs->B_blinding);
  ge_add(&c, &a, &b);
  ristretto_encode(cs->V[cs->m], &c);
  if (out) memcpy(out, cs->V[cs->m], 32);
  cs->v[cs->m] = *v; cs->v_blinding[cs->m] = *v_blinding;
  merlin_append_point(&cs->tr, "V", cs->V[cs->m]);
  r1cs_var r = {VAR_COMMITTED, (uint32_t)cs->m};
  cs->m++;
  return r;
}

r1cs_var r1cs_verifier_commit(r1cs_cs *cs, const uint8_t commitment[32]) {
  grow_m(cs);
  memcpy(cs->V[cs->m], commitment, 32);
  merlin_append_point(&cs->tr, "V", commitment);
  r1cs_var r = {VAR_COMMITTED, (uint32_t)cs->m};
  cs->m++;
  return r;
}

static size_t new_multiplier(r1cs_cs *cs) {
  if (cs->is_prover && cs->num_vars == cs->cap_vars) {
    cs->cap_vars = cs->cap_vars ? 2 * cs->cap_vars : 64;
    cs->aL = realloc(cs->aL, sizeof(sc) * cs->cap_vars);
    cs->aR = realloc(cs->aR, sizeof(sc) * cs->cap_vars);
    cs->aO = realloc(cs->aO, sizeof(sc) * cs->cap_vars);
  }
  return cs->num_vars++;
}

static void eval_lc(const r1cs_cs *cs, sc *out, const r1cs_lc *l) {
  sc acc, t, one;
  sc_from_u64(&acc, 0); sc_from_u64(&one, 1);
  for (size_t i = 0; i < l->n; ++i) {
    const sc *val = &one;
    switch (l->t[i].v.kind) {
      case VAR_COMMITTED: val = &cs->v[l->t[i].v.idx]; break;
      case VAR_MUL_LEFT: val = &cs->aL[l->t[i].v.idx]; break;
      case VAR_MUL_RIGHT: val = &cs->aR[l->t[i].v.idx]; break;
      case VAR_MUL_OUT: val = &cs->aO[l->t[i].v.idx]; break;
      default: break;
    }
    sc_mul(&t, &l->t[i].c, val);
    sc_add(&acc, &acc, &t);
  }
  *out = acc;
}

/* takes ownership of l */
void r1cs_constrain(r1cs_cs *cs, r1cs_lc *l) {
  if (cs->n_cons == cs->cap_cons) {
    cs->cap_cons = cs->cap_cons ? 2 * cs->cap_cons : 64;
    cs->cons = realloc(cs->cons, sizeof(r1cs_lc) * cs->cap_cons);
  }
  cs->cons[cs->n_cons++] = *l;
  lc_init(l);
}

/* multiply(left, right) -> (l, r, o); consumes both combinations */
void r1cs_multiply(r1cs_cs *cs, r1cs_lc *left, r1cs_lc *right, r1cs_var out[3]) {
  size_t i = new_multiplier(cs);
  if (cs->is_prover) {
    eval_lc(cs, &cs->aL[i], left);
    eval_lc(cs, &cs->aR[i], right);
    sc_mul(&cs->aO[i], &cs->aL[i], &cs->aR[i]);
  }
  r1cs_var l = {VAR_MUL_LEFT, (uint32_t)i}, r = {VAR_MUL_RIGHT, (uint32_t)i}, o = {VAR_MUL_OUT, (uint32_t)i};
  lc_push_neg_u64(left, l, 1);
  lc_push_neg_u64(right, r, 1);
  r1cs_constrain(cs, left);
  r1cs_constrain(cs, right);
  out[0] = l; out[1] = r; out[2] = o;
}

r1cs_var r1cs_allocate(r1cs_cs *cs, const sc *assignment) {
  if (cs->pending < 0) {
    size_t i = new_multiplier(cs);
    cs->pending = (long)i;
    if (cs->is_prover) { cs->aL[i] = *assignment; sc_from_u64(&cs->aR[i], 0); sc_from_u64(&cs->aO[i], 0); }
    r1cs_var v = {VAR_MUL_LEFT, (uint32_t)i};
    return v;
  }
  size_t i = (size_t)cs->pending;
  cs->pending = -1;
  if (cs->is_prover) { cs->aR[i] = *assignment; sc_mul(&cs->aO[i], &cs->aL[i], &cs->aR[i]); }
  r1cs_var v = {VAR_MUL_RIGHT, (uint32_t)i};
  return v;
}

void r1cs_allocate_multiplier(r1cs_cs *cs, const sc *l, const sc *r, r1cs_var out[3]) {
  size_t i = new_multiplier(cs);
  if (cs->is_prover) { cs->aL[i] = *l; cs->aR[i] = *r; sc_mul(&cs->aO[i], l, r); }
  out[0].kind = VAR_MUL_LEFT; out[1].kind = VAR_MUL_RIGHT; out[2].kind = VAR_MUL_OUT;
  out[0].idx = out[1].idx = out[2].idx = (uint32_t)i;
}

int r1cs_is_prover(const r1cs_cs *cs) { return cs->is_prover; }
size_t r1cs_num_multipliers(const r1cs_cs *cs) { return cs->num_vars; }
size_t r1cs_num_constraints(const r1cs_cs *cs) { return cs->n_cons; }
size_t r1cs_num_commitments(const r1cs_cs *cs) { return cs->m; }

int r1cs_specify_randomized_constraints(r1cs_cs *cs, r1cs_rand_fn fn, void *ud) {
  if (cs->phase == 2) return fn(cs, ud);  /* RandomizingProver/Verifier run nested callbacks at once */
  if (cs->n_def == cs->cap_def) {
    cs->cap_def = cs->cap_def ? 2 * cs->cap_def : 8;
    cs->deferred = realloc(cs->deferred, sizeof(*cs->deferred) * cs->cap_def);
  }
  cs->deferred[cs->n_def].fn = fn; cs->deferred[cs->n_def].ud = ud; cs->n_def++;
  return 0;
}

void r1cs_challenge_scalar(r1cs_cs *cs, const char *label, sc *out) { draw_challenge(cs, label, out); }

static int create_randomized_constraints(r1cs_cs *cs) {
  cs->pending = -1;
  if (cs->n_def == 0) {
    merlin_append_message(&cs->tr, "dom-sep", (const uint8_t *)"r1cs-1phase", 11);
    return 0;
  }
  merlin_append_message(&cs->tr, "dom-sep", (const uint8_t *)"r1cs-2phase", 11);
  cs->phase = 2;
  for (size_t i = 0; i < cs->n_def; ++i) {
    int rc = cs->deferred[i].fn(cs, cs->deferred[i].ud);
    if (rc) return rc;
  }
  cs->n_def = 0;
  return 0;
}

/* wL, wR, wO (num_vars each), wV (m), wc from the constraints weighted by z, z^2, ... */
static void flattened_constraints(const r1cs_cs *cs, const sc *z, sc *wL, sc *wR, sc *wO, sc *wV, sc *wc) {
  size_t n = cs->num_vars;
  sc zero; sc_from_u64(&zero, 0);
  for (size_t i = 0; i < n; ++i) wL[i] = wR[i] = wO[i] = zero;
  for (size_t i = 0; i < cs->m; ++i) wV[i] = zero;
  *wc = zero;
  sc exp_z = *z, t;
  for (size_t q = 0; q < cs->n_cons; ++q) {
    const r1cs_lc *l = &cs->cons[q];
    for (size_t k = 0; k < l->n; ++k) {
      sc_mul(&t, &exp_z, &l->t[k].c);
      uint32_t i = l->t[k].v.idx;
      switch (l->t[k].v.kind) {
        case VAR_MUL_LEFT: sc_add(&wL[i], &wL[i], &t); break;
        case VAR_MUL_RIGHT: sc_add(&wR[i], &wR[i], &t); break;
        case VAR_MUL_OUT: sc_add(&wO[i], &wO[i], &t); break;
        case VAR_COMMITTED: sc_sub(&wV[i], &wV[i], &t); break;
        default: sc_sub(wc, wc, &t); break;
      }
    }
    sc_mul(&exp_z, &exp_z, z);
  }
}

static size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }
static int lg2(size_t x) { int k = 0; while (((size_t)1 << k) < x) ++k; return k; }

size_t r1cs_proof_size(size_t padded_n) { return 1 + 32 * (14 + 2 * (size_t)lg2(padded_n) + 2); }

/* ---- prover ------------------------------------------------------------------ */
static void rng_scalar(merlin_transcript *rng, sc *out) {
  uint8_t b[64];
  merlin_rng_fill(rng, b, 64);
  sc_from_bytes_wide(out, b);
}

static void commit_vec(uint8_t out[32], const sc *blind, const ge *Bb, const sc *a, const ge *G, size_t na,
                       const sc *b, const ge *H, size_t nb) {
  size_t n = 1 + na + nb;
  sc *s = malloc(sizeof(sc) * n);
  ge *p = malloc(sizeof(ge) * n);
  s[0] = *blind; p[0] = *Bb;
  for (size_t i = 0; i < na; ++i) { s[1 + i] = a[i]; p[1 + i] = G[i]; }
  for (size_t i = 0; i < nb; ++i) { s[1 + na + i] = b[i]; p[1 + na + i] = H[i]; }
  ge r;
  ge_msm_vartime(&r, s, p, n);
  ristretto_encode(out, &r);
  free(s); free(p);
}

static void pedersen_commit(uint8_t out[32], const r1cs_cs *cs, const sc *v, const sc *blind) {
  sc s[2] = {*v, *blind};
  ge p[2] = {cs->B, cs->B_blinding}, r;
  ge_msm_vartime(&r, s, p, 2);
  ristretto_encode(out, &r);
}

static void inner(sc *out, const sc *a, const sc *b, size_t n) {
  sc acc, t; sc_from_u64(&acc, 0);
  for (size_t i = 0; i < n; ++i) { sc_mul(&t, &a[i], &b[i]); sc_add(&acc, &acc, &t); }
  *out = acc;
}

int r1cs_prove(r1cs_cs *cs, const uint8_t rng_seed[32], uint8_t *proof, size_t proof_cap, size_t *proof_len) {
  if (!cs->is_prover) return -1;
  merlin_transcript *tr = &cs->tr;
  merlin_append_u64(tr, "m", cs->m);
  merlin_transcript rng = *tr;
  for (size_t i = 0; i < cs->m; ++i) {
    uint8_t b[32]; sc_to_bytes(b, &cs->v_blinding[i]);
    merlin_rekey_with_witness(&rng, "v_blinding", b, 32);
  }
  merlin_finalize_rng(&rng, rng_seed);

  size_t n1 = cs->num_vars;
  gens_ensure(next_pow2(n1 ? n1 : 1));
  sc i_bl1, o_bl1, s_bl1;
  rng_scalar(&rng, &i_bl1); rng_scalar(&rng, &o_bl1); rng_scalar(&rng, &s_bl1);
  sc *sL1 = malloc(sizeof(sc) * (n1 + 1)), *sR1 = malloc(sizeof(sc) * (n1 + 1));
  for (size_t i = 0; i < n1; ++i) rng_scalar(&rng, &sL1[i]);
  for (size_t i = 0; i < n1; ++i) rng_scalar(&rng, &sR1[i]);
  uint8_t A_I1[32], A_O1[32], S1[32];
  commit_vec(A_I1, &i_bl1, &cs->B_blinding, cs->aL, g_cacheG, n1, cs->aR, g_cacheH, n1);
  commit_vec(A_O1, &o_bl1, &cs->B_blinding, cs->aO, g_cacheG, n1, NULL, NULL, 0);
  commit_vec(S1, &s_bl1, &cs->B_blinding, sL1, g_cacheG, n1, sR1, g_cacheH, n1);
  merlin_append_point(tr, "A_I1", A_I1);
  merlin_append_point(tr, "A_O1", A_O1);
  merlin_append_point(tr, "S1", S1);

  if (create_randomized_constraints(cs)) { free(sL1); free(sR1); return -2; }

  size_t n = cs->num_vars, n2 = n - n1, pn = next_pow2(n ? n : 1), pad = pn - n;
  int k = lg2(pn);
  gens_ensure(pn);
  const ge *G = g_cacheG, *H = g_cacheH;
  if (proof_cap < r1cs_proof_size(pn)) { free(sL1); free(sR1); return -3; }

  sc i_bl2, o_bl2, s_bl2, zero, one;
  sc_from_u64(&zero, 0); sc_from_u64(&one, 1);
  i_bl2 = o_bl2 = s_bl2 = zero;
  sc *sL = malloc(sizeof(sc) * (n + 1)), *sR = malloc(sizeof(sc) * (n + 1));
  memcpy(sL, sL1, sizeof(sc) * n1); memcpy(sR, sR1, sizeof(sc) * n1);
  uint8_t A_I2[32], A_O2[32], S2[32];
  memset(A_I2, 0, 32); memset(A_O2, 0, 32); memset(S2, 0, 32); /* identity */
  if (n2 > 0) {
    rng_scalar(&rng, &i_bl2); rng_scalar(&rng, &o_bl2); rng_scalar(&rng, &s_bl2);
    for (size_t i = n1; i < n; ++i) rng_scalar(&rng, &sL[i]);
    for (size_t i = n1; i < n; ++i) rng_scalar(&rng, &sR[i]);
    commit_vec(A_I2, &i_bl2, &cs->B_blinding, cs->aL + n1, G + n1, n2, cs->aR + n1, H + n1, n2);
    commit_vec(A_O2, &o_bl2, &cs->B_blinding, cs->aO + n1, G + n1, n2, NULL, NULL, 0);
    commit_vec(S2, &s_bl2, &cs->B_blinding, sL + n1, G + n1, n2, sR + n1, H + n1, n2);
  }
  merlin_append_point(tr, "A_I2", A_I2);
  merlin_append_point(tr, "A_O2", A_O2);
  merlin_append_point(tr, "S2", S2);

  sc y, z;
  merlin_challenge_scalar(tr, "y", &y);
  merlin_challenge_scalar(tr, "z", &z);
  sc *wL = malloc(sizeof(sc) * (n + 1)), *wR = malloc(sizeof(sc) * (n + 1)), *wO = malloc(sizeof(sc) * (n + 1));
  sc *wV = malloc(sizeof(sc) * (cs->m + 1)), wc;
  flattened_constraints(cs, &z, wL, wR, wO, wV, &wc);

  /* l(x) = l1 x + l2 x^2 + l3 x^3,  r(x) = r0 + r1 x + r3 x^3 */
  sc *l1 = malloc(sizeof(sc) * pn), *l2 = malloc(sizeof(sc) * pn), *l3 = malloc(sizeof(sc) * pn);
  sc *r0 = malloc(sizeof(sc) * pn), *r1 = malloc(sizeof(sc) * pn), *r3 = malloc(sizeof(sc) * pn);
  sc *yinv_pow = malloc(sizeof(sc) * pn);
  sc y_inv, exp_y = one, t;
  sc_invert(&y_inv, &y);
  yinv_pow[0] = one;
  for (size_t i = 1; i < pn; ++i) sc_mul(&yinv_pow[i], &yinv_pow[i - 1], &y_inv);
  for (size_t i = 0; i < n; ++i) {
    sc_mul(&t, &yinv_pow[i], &wR[i]); sc_add(&l1[i], &cs->aL[i], &t);
    l2[i] = cs->aO[i];
    l3[i] = sL[i];
    sc_sub(&r0[i], &wO[i], &exp_y);
    sc_mul(&t, &exp_y, &cs->aR[i]); sc_add(&r1[i], &t, &wL[i]);
    sc_mul(&r3[i], &exp_y, &sR[i]);
    sc_mul(&exp_y, &exp_y, &y);
  }
  sc t1, t2, t3, t4, t5, t6, u1, u2;
  inner(&t1, l1, r0, n);
  inner(&u1, l1, r1, n); inner(&u2, l2, r0, n); sc_add(&t2, &u1, &u2);
  inner(&u1, l2, r1, n); inner(&u2, l3, r0, n); sc_add(&t3, &u1, &u2);
  inner(&u1, l1, r3, n); inner(&u2, l3, r1, n); sc_add(&t4, &u1, &u2);
  inner(&t5, l2, r3, n);
  inner(&t6, l3, r3, n);
  sc tb1, tb3, tb4, tb5, tb6;
  rng_scalar(&rng, &tb1); rng_scalar(&rng, &tb3); rng_scalar(&rng, &tb4); rng_scalar(&rng, &tb5); rng_scalar(&rng, &tb6);
  uint8_t T1[32], T3[32], T4[32], T5[32], T6[32];
  pedersen_commit(T1, cs, &t1, &tb1); pedersen_commit(T3, cs, &t3, &tb3); pedersen_commit(T4, cs, &t4, &tb4);
  pedersen_commit(T5, cs, &t5, &tb5); pedersen_commit(T6, cs, &t6, &tb6);
  merlin_append_point(tr, "T_1", T1); merlin_append_point(tr, "T_3", T3); merlin_append_point(tr, "T_4", T4);
  merlin_append_point(tr, "T_5", T5); merlin_append_point(tr, "T_6", T6);

  sc u, x;
  merlin_challenge_scalar(tr, "u", &u);
  merlin_challenge_scalar(tr, "x", &x);
  sc tb2 = zero;
  for (size_t i = 0; i < cs->m; ++i) { sc_mul(&t, &wV[i], &cs->v_blinding[i]); sc_add(&tb2, &tb2, &t); }
  /* evaluate t(x), t_blinding(x) */
  sc xp[7]; xp[0] = one;
  for (int i = 1; i <= 6; ++i) sc_mul(&xp[i], &xp[i - 1], &x);
  sc t_x = zero, t_x_bl = zero;
  const sc *tc[7] = {NULL, &t1, &t2, &t3, &t4, &t5, &t6}, *bc[7] = {NULL, &tb1, &tb2, &tb3, &tb4, &tb5, &tb6};
  for (int i = 1; i <= 6; ++i) {
    sc_mul(&t, tc[i], &xp[i]); sc_add(&t_x, &t_x, &t);
    sc_mul(&t, bc[i], &xp[i]); sc_add(&t_x_bl, &t_x_bl, &t);
  }
  sc *lv = malloc(sizeof(sc) * pn), *rv = malloc(sizeof(sc) * pn);
  for (size_t i = 0; i < n; ++i) {
    sc a1, a2, a3;
    sc_mul(&a1, &l1[i], &xp[1]); sc_mul(&a2, &l2[i], &xp[2]); sc_mul(&a3, &l3[i], &xp[3]);
    sc_add(&lv[i], &a1, &a2); sc_add(&lv[i], &lv[i], &a3);
    sc_mul(&a1, &r1[i], &xp[1]); sc_mul(&a3, &r3[i], &xp[3]);
    sc_add(&rv[i], &r0[i], &a1); sc_add(&rv[i], &rv[i], &a3);
  }
  for (size_t i = n; i < pn; ++i) { lv[i] = zero; sc_neg(&rv[i], &exp_y); sc_mul(&exp_y, &exp_y, &y); }

  sc i_bl, o_bl, s_bl, e_bl;
  sc_mul(&t, &u, &i_bl2); sc_add(&i_bl, &i_bl1, &t);
  sc_mul(&t, &u, &o_bl2); sc_add(&o_bl, &o_bl1, &t);
  sc_mul(&t, &u, &s_bl2); sc_add(&s_bl, &s_bl1, &t);
  sc_mul(&t, &x, &s_bl); sc_add(&t, &t, &o_bl); sc_mul(&t, &t, &x); sc_add(&t, &t, &i_bl); sc_mul(&e_bl, &t, &x);
  merlin_append_scalar(tr, "t_x", &t_x);
  merlin_append_scalar(tr, "t_x_blinding", &t_x_bl);
  merlin_append_scalar(tr, "e_blinding", &e_bl);
  sc w;
  merlin_challenge_scalar(tr, "w", &w);
  ge Q;
  ge_scalarmult(&Q, &w, &cs->B);

  /* inner-product argument; folded generators are tracked as coefficient
   * vectors over the original ones, so every L, R is one MSM of size pn + 1 */
  sc *cG = malloc(sizeof(sc) * pn), *cH = malloc(sizeof(sc) * pn);
  for (size_t i = 0; i < pn; ++i) {
    cG[i] = (i < n1) ? one : u;
    sc_mul(&cH[i], &yinv_pow[i], &cG[i]);
  }
  uint8_t *p = proof;
  *p++ = 1; /* two-phase wire format */
  memcpy(p, A_I1, 32); p += 32; memcpy(p, A_O1, 32); p += 32; memcpy(p, S1, 32); p += 32;
  memcpy(p, A_I2, 32); p += 32; memcpy(p, A_O2, 32); p += 32; memcpy(p, S2, 32); p += 32;
  memcpy(p, T1, 32); p += 32; memcpy(p, T3, 32); p += 32; memcpy(p, T4, 32); p += 32;
  memcpy(p, T5, 32); p += 32; memcpy(p, T6, 32); p += 32;
  sc_to_bytes(p, &t_x); p += 32; sc_to_bytes(p, &t_x_bl); p += 32; sc_to_bytes(p, &e_bl); p += 32;

  merlin_append_message(tr, "dom-sep", (const uint8_t *)"ipp v1", 6);
  merlin_append_u64(tr, "n", pn);
  sc *ms = malloc(sizeof(sc) * (pn + 1));
  ge *mp = malloc(sizeof(ge) * (pn + 1));
  size_t len = pn;
  for (int round = 0; round < k; ++round) {
    size_t half = len / 2;
    sc cL, cR;
    inner(&cL, lv, rv + half, half);
    inner(&cR, lv + half, rv, half);
    for (int side = 0; side < 2; ++side) { /* 0: L, 1: R */
      size_t cnt = 0;
      for (size_t idx = 0; idx < pn; ++idx) {
        size_t j = idx % len;
        int hi = j >= half;
        size_t jj = hi ? j - half : j;
        if (side == 0) {
          if (hi) { sc_mul(&ms[cnt], &lv[jj], &cG[idx]); mp[cnt++] = G[idx]; }          /* a_L * G_R */
          else { sc_mul(&ms[cnt], &rv[half + jj], &cH[idx]); mp[cnt++] = H[idx]; }      /* b_R * H_L */
        } else {
          if (!hi) { sc_mul(&ms[cnt], &lv[half + jj], &cG[idx]); mp[cnt++] = G[idx]; }  /* a_R * G_L */
          else { sc_mul(&ms[cnt], &rv[jj], &cH[idx]); mp[cnt++] = H[idx]; }             /* b_L * H_R */
        }
      }
      ms[cnt] = side ? cR : cL; mp[cnt++] = Q;
      ge r;
      ge_msm_vartime(&r, ms, mp, cnt);
      ristretto_encode(p, &r);
      merlin_append_point(tr, side ? "R" : "L", p);
      p += 32;
    }
    sc uu, uu_inv;
    merlin_challenge_scalar(tr, "u", &uu);
    sc_invert(&uu_inv, &uu);
    for (size_t j = 0; j < half; ++j) {
      sc a, b;
      sc_mul(&a, &lv[j], &uu); sc_mul(&b, &lv[half + j], &uu_inv); sc_add(&lv[j], &a, &b);
      sc_mul(&a, &rv[j], &uu_inv); sc_mul(&b, &rv[half + j], &uu); sc_add(&rv[j], &a, &b);
    }
    for (size_t idx = 0; idx < pn; ++idx) {
      int hi = (idx % len) >= half;
      sc_mul(&cG[idx], &cG[idx], hi ? &uu : &uu_inv);
      sc_mul(&cH[idx], &cH[idx], hi ? &uu_inv : &uu);
    }
    len = half;
  }
  sc_to_bytes(p, &lv[0]); p += 32;
  sc_to_bytes(p, &rv[0]); p += 32;
  *proof_len = (size_t)(p - proof);

  free(sL1); free(sR1); free(sL); free(sR); free(wL); free(wR); free(wO); free(wV);
  free(l1); free(l2); free(l3); free(r0); free(r1); free(r3); free(yinv_pow); free(lv); free(rv);
  free(cG); free(cH); free(ms); free(mp);
  (void)pad;
  return 0;
}

/* ---- verifier ------------------------------------------------------------------ */
/* Builds the terms of the single verification MSM (dalek `mega_check`) in the
 * order  [A_I1 A_O1 S1 A_I2 A_O2 S2 | V.. | T_1 T_3 T_4 T_5 T_6 | L.. R..]  (dynamic)
 * and    [B, B_blinding, G_0..G_{pn-1}, H_0..H_{pn-1}]                       (static).
 * r_bytes: the verifier's random weight r (64 uniform bytes).  Returns 0, or <0
 * when the proof is malformed (wrong size / non-canonical scalar / identity
 * where `validate_and_append_point` forbids it). */
int r1cs_verify_prepare(r1cs_cs *cs, const uint8_t *proof, size_t proof_len, const uint8_t r_bytes[64],
                        r1cs_msm *out) {
  if (cs->is_prover) return -1;
  merlin_transcript *tr = &cs->tr;
  memset(out, 0, sizeof *out);
  /* the one-phase wire format (version byte 0: A_I2, A_O2, S2 left out, 13 + 2k elements) is the two-phase one with
   * the identity in their place (upstream R1CSProof::from_bytes) */
  uint8_t expanded[1 + 32 * (16 + 2 * 32)];
  if (proof_len >= 1 + 32 * 13 && proof[0] == 0 && (proof_len - 1) % 32 == 0 && proof_len + 96 <= sizeof expanded) {
    expanded[0] = 1;
    memcpy(expanded + 1, proof + 1, 96);
    memset(expanded + 97, 0, 96);
    memcpy(expanded + 193, proof + 97, proof_len - 97);
    proof = expanded;
    proof_len += 96;
  }
  if (proof_len < 1 + 32 * 16 || proof[0] != 1) return -2;
  if ((proof_len - 1) % 32) return -2;
  size_t words = (proof_len - 1) / 32;
  if (words < 16 || (words - 16) % 2) return -2;
  size_t k = (words - 16) / 2;
  if (k >= 32) return -2;
  const uint8_t *pt = proof + 1;            /* 11 points */
  const uint8_t *scb = pt + 32 * 11;        /* t_x, t_x_blinding, e_blinding */
  const uint8_t *lr = scb + 32 * 3;         /* L_0 R_0 L_1 R_1 ... */
  const uint8_t *ab = lr + 64 * k;
  sc t_x, t_x_bl, e_bl, a, b;
  if (!sc_from_canonical_bytes(&t_x, scb) || !sc_from_canonical_bytes(&t_x_bl, scb + 32) ||
      !sc_from_canonical_bytes(&e_bl, scb + 64) || !sc_from_canonical_bytes(&a, ab) ||
      !sc_from_canonical_bytes(&b, ab + 32))
    return -3;
  static const uint8_t ident[32] = {0};
#define VALIDATE(ptr) do { if (memcmp((ptr), ident, 32) == 0) return -4; } while (0)

  merlin_append_u64(tr, "m", cs->m);
  size_t n1 = cs->num_vars;
  VALIDATE(pt); VALIDATE(pt + 32); VALIDATE(pt + 64);
  merlin_append_point(tr, "A_I1", pt);
  merlin_append_point(tr, "A_O1", pt + 32);
  merlin_append_point(tr, "S1", pt + 64);
  if (create_randomized_constraints(cs)) return -5;
  size_t n = cs->num_vars, n2 = n - n1, pn = next_pow2(n ? n : 1), pad = pn - n;
  if ((size_t)1 << k != pn) return -6;
  merlin_append_point(tr, "A_I2", pt + 96);
  merlin_append_point(tr, "A_O2", pt + 128);
  merlin_append_point(tr, "S2", pt + 160);
  sc y, z, u, x, w;
  draw_challenge(cs, "y", &y);
  draw_challenge(cs, "z", &z);
  for (int i = 6; i < 11; ++i) VALIDATE(pt + 32 * i);
  merlin_append_point(tr, "T_1", pt + 192); merlin_append_point(tr, "T_3", pt + 224);
  merlin_append_point(tr, "T_4", pt + 256); merlin_append_point(tr, "T_5", pt + 288);
  merlin_append_point(tr, "T_6", pt + 320);
  draw_challenge(cs, "u", &u);
  draw_challenge(cs, "x", &x);
  merlin_append_scalar(tr, "t_x", &t_x);
  merlin_append_scalar(tr, "t_x_blinding", &t_x_bl);
  merlin_append_scalar(tr, "e_blinding", &e_bl);
  draw_challenge(cs, "w", &w);

  sc *wL = malloc(sizeof(sc) * (n + 1)), *wR = malloc(sizeof(sc) * (n + 1)), *wO = malloc(sizeof(sc) * (n + 1));
  sc *wV = malloc(sizeof(sc) * (cs->m + 1)), wc;
  flattened_constraints(cs, &z, wL, wR, wO, wV, &wc);

  /* IPA verification scalars */
  merlin_append_message(tr, "dom-sep", (const uint8_t *)"ipp v1", 6);
  merlin_append_u64(tr, "n", pn);
  sc *ch = malloc(sizeof(sc) * (k + 1)), *ch_inv = malloc(sizeof(sc) * (k + 1));
  int rc = 0;
  for (size_t j = 0; j < k; ++j) {
    if (memcmp(lr + 64 * j, ident, 32) == 0 || memcmp(lr + 64 * j + 32, ident, 32) == 0) rc = -4;
    merlin_append_point(tr, "L", lr + 64 * j);
    merlin_append_point(tr, "R", lr + 64 * j + 32);
    draw_challenge(cs, "u", &ch[j]);
  }
  if (rc) { free(wL); free(wR); free(wO); free(wV); free(ch); free(ch_inv); return rc; }
  sc one, zero, allinv;
  sc_from_u64(&one, 1); sc_from_u64(&zero, 0);
  allinv = one;
  for (size_t j = 0; j < k; ++j) { sc_invert(&ch_inv[j], &ch[j]); sc_mul(&allinv, &allinv, &ch_inv[j]); }
  sc *u_sq = malloc(sizeof(sc) * (k + 1)), *u_inv_sq = malloc(sizeof(sc) * (k + 1));
  for (size_t j = 0; j < k; ++j) { sc_mul(&u_sq[j], &ch[j], &ch[j]); sc_mul(&u_inv_sq[j], &ch_inv[j], &ch_inv[j]); }
  sc *s = malloc(sizeof(sc) * pn);
  s[0] = allinv;
  for (size_t i = 1; i < pn; ++i) {
    int lg_i = 0;
    while (((size_t)2 << lg_i) <= i) ++lg_i;
    size_t kk = (size_t)1 << lg_i;
    sc_mul(&s[i], &s[i - kk], &u_sq[(k - 1) - (size_t)lg_i]);
  }

  sc y_inv, t, t2;
  sc_invert(&y_inv, &y);
  sc *yinv_pow = malloc(sizeof(sc) * pn), *yneg_wR = malloc(sizeof(sc) * pn);
  yinv_pow[0] = one;
  for (size_t i = 1; i < pn; ++i) sc_mul(&yinv_pow[i], &yinv_pow[i - 1], &y_inv);
  for (size_t i = 0; i < pn; ++i) {
    if (i < n) sc_mul(&yneg_wR[i], &wR[i], &yinv_pow[i]); else yneg_wR[i] = zero;
  }
  sc delta;
  inner(&delta, yneg_wR, wL, n);

  sc r;
  sc_from_bytes_wide(&r, r_bytes);
  sc xx, xxx, rxx;
  sc_mul(&xx, &x, &x); sc_mul(&xxx, &xx, &x); sc_mul(&rxx, &r, &xx);

  size_t m = cs->m;
  out->n_dyn = 6 + m + 5 + 2 * k;
  out->n_static = 2 + 2 * pn;
  out->padded_n = pn;
  out->dyn_scalars = malloc(32 * out->n_dyn);
  out->dyn_points = malloc(32 * out->n_dyn);
  out->static_scalars = malloc(32 * out->n_static);
  uint8_t *ds = out->dyn_scalars, *dp = out->dyn_points, *ss = out->static_scalars;
#define PUSH_DYN(scalar, point) do { sc_to_bytes(ds, (scalar)); ds += 32; memcpy(dp, (point), 32); dp += 32; } while (0)
  sc ux, uxx, uxxx;
  sc_mul(&ux, &u, &x); sc_mul(&uxx, &u, &xx); sc_mul(&uxxx, &u, &xxx);
  PUSH_DYN(&x, pt); PUSH_DYN(&xx, pt + 32); PUSH_DYN(&xxx, pt + 64);
  PUSH_DYN(&ux, pt + 96); PUSH_DYN(&uxx, pt + 128); PUSH_DYN(&uxxx, pt + 160);
  for (size_t i = 0; i < m; ++i) { sc_mul(&t, &wV[i], &rxx); PUSH_DYN(&t, cs->V[i]); }
  sc T_s[5];
  sc_mul(&T_s[0], &r, &x);            /* r x      T_1 */
  sc_mul(&T_s[1], &rxx, &x);          /* r x^3    T_3 */
  sc_mul(&T_s[2], &rxx, &xx);         /* r x^4    T_4 */
  sc_mul(&T_s[3], &rxx, &xxx);        /* r x^5    T_5 */
  sc_mul(&T_s[4], &T_s[2], &xx);      /* r x^6    T_6 */
  for (int i = 0; i < 5; ++i) PUSH_DYN(&T_s[i], pt + 192 + 32 * i);
  for (size_t j = 0; j < k; ++j) PUSH_DYN(&u_sq[j], lr + 64 * j);
  for (size_t j = 0; j < k; ++j) PUSH_DYN(&u_inv_sq[j], lr + 64 * j + 32);
  /* B: w (t_x - a b) + r (xx (wc + delta) - t_x);  B_blinding: -e_blinding - r t_x_blinding */
  sc ab_, sB, sBb;
  sc_mul(&ab_, &a, &b); sc_sub(&t, &t_x, &ab_); sc_mul(&sB, &w, &t);
  sc_add(&t, &wc, &delta); sc_mul(&t, &t, &xx); sc_sub(&t, &t, &t_x); sc_mul(&t, &t, &r); sc_add(&sB, &sB, &t);
  sc_mul(&t, &r, &t_x_bl); sc_add(&t, &t, &e_bl); sc_neg(&sBb, &t);
  sc_to_bytes(ss, &sB); ss += 32;
  sc_to_bytes(ss, &sBb); ss += 32;
  /* g_i = u_or_1 (x yneg_wR_i - a s_i);  h_i = u_or_1 (y^-i (x wL_i + wO_i - b s_{pn-1-i}) - 1) */
  for (size_t i = 0; i < pn; ++i) {
    sc_mul(&t, &x, &yneg_wR[i]); sc_mul(&t2, &a, &s[i]); sc_sub(&t, &t, &t2);
    if (i >= n1) sc_mul(&t, &t, &u);
    sc_to_bytes(ss, &t); ss += 32;
  }
  for (size_t i = 0; i < pn; ++i) {
    sc wl = i < n ? wL[i] : zero, wo = i < n ? wO[i] : zero;
    sc_mul(&t, &x, &wl); sc_add(&t, &t, &wo); sc_mul(&t2, &b, &s[pn - 1 - i]); sc_sub(&t, &t, &t2);
    sc_mul(&t, &t, &yinv_pow[i]); sc_sub(&t, &t, &one);
    if (i >= n1) sc_mul(&t, &t, &u);
    sc_to_bytes(ss, &t); ss += 32;
  }
  free(wL); free(wR); free(wO); free(wV); free(ch); free(ch_inv); free(u_sq); free(u_inv_sq); free(s);
  free(yinv_pow); free(yneg_wR);
  (void)pad; (void)n2;
  return 0;
}

void r1cs_msm_free(r1cs_msm *m) {
  free(m->dyn_scalars); free(m->dyn_points); free(m->static_scalars);
  memset(m, 0, sizeof *m);
}

/* full CPU verification: prepare + MSM == identity (the reference's Verifier::verify) */
int r1cs_verify(r1cs_cs *cs, const uint8_t *proof, size_t proof_len, const uint8_t r_bytes[64]) {
  r1cs_msm m;
  int rc = r1cs_verify_prepare(cs, proof, proof_len, r_bytes, &m);
  if (rc) return 0;
  size_t n = m.n_dyn + m.n_static;
  sc *s = malloc(sizeof(sc) * n);
  ge *p = malloc(sizeof(ge) * n);
  int ok = 1;
  for (size_t i = 0; i < m.n_dyn && ok; ++i) {
    sc_from_bytes_mod_order(&s[i], m.dyn_scalars + 32 * i);
    ok = ristretto_decode(&p[i], m.dyn_points + 32 * i);
  }
  if (ok) {
    gens_ensure(m.padded_n);
    size_t o = m.n_dyn;
    for (size_t i = 0; i < m.n_static; ++i) sc_from_bytes_mod_order(&s[o + i], m.static_scalars + 32 * i);
    p[o] = cs->B; p[o + 1] = cs->B_blinding;
    for (size_t i = 0; i < m.padded_n; ++i) { p[o + 2 + i] = g_cacheG[i]; p[o + 2 + m.padded_n + i] = g_cacheH[i]; }
    ge r;
    ge_msm_vartime(&r, s, p, n);
    ok = ge_is_identity(&r);
  }
  free(s); free(p);
  r1cs_msm_free(&m);
  return ok;
}
