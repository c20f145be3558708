/* merlin.c -- STROBE-128 (the AD / meta-AD / PRF / KEY subset) and Merlin v1.0
 * transcripts, plus the Bulletproofs generator chains.
 *
 * TEST INFRASTRUCTURE (see oracle.h).  Restates the `merlin` crate and
 * bulletproofs' `TranscriptProtocol`, `PedersenGens`, `BulletproofGens`
 * (sources not mounted) from the merlin.cool
 * transcript specification, STROBE v1.0.2 sec 5-6 and FIPS 202.  Pinned by the
 * public Merlin "test protocol" known answer and dalek's published Pedersen
 * blinding base (tests/golden/transcript.json).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define STROBE_R 166
#define FLAG_I 1
#define FLAG_A 2
#define FLAG_C 4
#define FLAG_T 8
#define FLAG_M 16
#define FLAG_K 32

static void strobe_f(strobe128 *s) {
  uint64_t st[25];
  for (int i = 0; i < 25; ++i) {
    st[i] = 0;
    for (int j = 7; j >= 0; --j) st[i] = (st[i] << 8) | s->st[8 * i + j];
  }
  keccak_f1600(st);
  for (int i = 0; i < 25; ++i)
    for (int j = 0; j < 8; ++j) s->st[8 * i + j] = (uint8_t)(st[i] >> (8 * j));
}

static void strobe_run_f(strobe128 *s) {
  s->st[s->pos] ^= s->pos_begin;
  s->st[s->pos + 1] ^= 0x04;
  s->st[STROBE_R + 1] ^= 0x80;
  strobe_f(s);
  s->pos = 0;
  s->pos_begin = 0;
}

static void strobe_absorb(strobe128 *s, const uint8_t *d, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    s->st[s->pos++] ^= d[i];
    if (s->pos == STROBE_R) strobe_run_f(s);
  }
}

static void strobe_overwrite(strobe128 *s, const uint8_t *d, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    s->st[s->pos++] = d[i];
    if (s->pos == STROBE_R) strobe_run_f(s);
  }
}

static void strobe_squeeze(strobe128 *s, uint8_t *d, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    d[i] = s->st[s->pos];
    s->st[s->pos++] = 0;
    if (s->pos == STROBE_R) strobe_run_f(s);
  }
}

static void strobe_begin_op(strobe128 *s, uint8_t flags, int more) {
  if (more) return; /* continuation of the current operation */
  uint8_t hdr[2] = {s->pos_begin, flags};
  s->pos_begin = (uint8_t)(s->pos + 1);
  s->cur_flags = flags;
  strobe_absorb(s, hdr, 2);
  if ((flags & (FLAG_C | FLAG_K)) && s->pos != 0) strobe_run_f(s);
}

static void strobe_meta_ad(strobe128 *s, const uint8_t *d, size_t n, int more) {
  strobe_begin_op(s, FLAG_M | FLAG_A, more);
  strobe_absorb(s, d, n);
}
static void strobe_ad(strobe128 *s, const uint8_t *d, size_t n, int more) {
  strobe_begin_op(s, FLAG_A, more);
  strobe_absorb(s, d, n);
}
static void strobe_prf(strobe128 *s, uint8_t *d, size_t n, int more) {
  strobe_begin_op(s, FLAG_I | FLAG_A | FLAG_C, more);
  strobe_squeeze(s, d, n);
}
static void strobe_key(strobe128 *s, const uint8_t *d, size_t n, int more) {
  strobe_begin_op(s, FLAG_A | FLAG_C, more);
  strobe_overwrite(s, d, n);
}

static void strobe_init(strobe128 *s, const uint8_t *label, size_t n) {
  static const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
  memset(s, 0, sizeof *s);
  memcpy(s->st, hdr, 6);
  memcpy(s->st + 6, "STROBEv1.0.2", 12);
  strobe_f(s);
  strobe_meta_ad(s, label, n, 0);
}

static void le32(uint8_t b[4], size_t n) {
  b[0] = (uint8_t)n; b[1] = (uint8_t)(n >> 8); b[2] = (uint8_t)(n >> 16); b[3] = (uint8_t)(n >> 24);
}

void merlin_init(merlin_transcript *t, const uint8_t *label, size_t len) {
  strobe_init(&t->s, (const uint8_t *)"Merlin v1.0", 11);
  merlin_append_message(t, "dom-sep", label, len);
}

void merlin_append_message(merlin_transcript *t, const char *label, const uint8_t *msg, size_t len) {
  uint8_t n[4];
  le32(n, len);
  strobe_meta_ad(&t->s, (const uint8_t *)label, strlen(label), 0);
  strobe_meta_ad(&t->s, n, 4, 1);
  strobe_ad(&t->s, msg, len, 0);
}

void merlin_append_u64(merlin_transcript *t, const char *label, uint64_t x) {
  uint8_t b[8];
  for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(x >> (8 * i));
  merlin_append_message(t, label, b, 8);
}

void merlin_challenge_bytes(merlin_transcript *t, const char *label, uint8_t *out, size_t len) {
  uint8_t n[4];
  le32(n, len);
  strobe_meta_ad(&t->s, (const uint8_t *)label, strlen(label), 0);
  strobe_meta_ad(&t->s, n, 4, 1);
  strobe_prf(&t->s, out, len, 0);
}

void merlin_append_scalar(merlin_transcript *t, const char *label, const sc *s) {
  uint8_t b[32];
  sc_to_bytes(b, s);
  merlin_append_message(t, label, b, 32);
}

void merlin_append_point(merlin_transcript *t, const char *label, const uint8_t p[32]) {
  merlin_append_message(t, label, p, 32);
}

void merlin_challenge_scalar(merlin_transcript *t, const char *label, sc *out) {
  uint8_t b[64];
  merlin_challenge_bytes(t, label, b, 64);
  sc_from_bytes_wide(out, b);
}

/* TranscriptRngBuilder::rekey_with_witness_bytes / finalize, TranscriptRng::fill_bytes.
 * Call on a *copy* of the transcript. */
void merlin_rekey_with_witness(merlin_transcript *t, const char *label, const uint8_t *w, size_t len) {
  uint8_t n[4];
  le32(n, len);
  strobe_meta_ad(&t->s, (const uint8_t *)label, strlen(label), 0);
  strobe_meta_ad(&t->s, n, 4, 1);
  strobe_key(&t->s, w, len, 0);
}

void merlin_finalize_rng(merlin_transcript *t, const uint8_t rng_seed[32]) {
  strobe_meta_ad(&t->s, (const uint8_t *)"rng", 3, 0);
  strobe_key(&t->s, rng_seed, 32, 0);
}

void merlin_rng_fill(merlin_transcript *t, uint8_t *out, size_t len) {
  uint8_t n[4];
  le32(n, len);
  strobe_meta_ad(&t->s, n, 4, 0);
  strobe_prf(&t->s, out, len, 0);
}

/* ---- generators ----------------------------------------------------------- */

/* PedersenGens::default(): B = ristretto basepoint, B_blinding =
 * from_uniform_bytes(SHA3-512(compress(B))). */
void pedersen_gens(ge *B, ge *B_blinding) {
  uint8_t enc[32], h[64];
  ge_basepoint(B);
  ristretto_encode(enc, B);
  sha3_512(h, enc, 32);
  ristretto_from_uniform_bytes(B_blinding, h);
}

/* GeneratorsChain: SHAKE256("GeneratorsChain" || which || LE32(party)), 64
 * bytes per generator through from_uniform_bytes. */
void bulletproof_gens_chain(ge *out, size_t n, char which, uint32_t party) {
  shake256_ctx c;
  uint8_t label[5] = {(uint8_t)which, (uint8_t)party, (uint8_t)(party >> 8), (uint8_t)(party >> 16),
                      (uint8_t)(party >> 24)};
  shake256_init(&c);
  shake256_absorb(&c, (const uint8_t *)"GeneratorsChain", 15);
  shake256_absorb(&c, label, 5);
  for (size_t i = 0; i < n; ++i) {
    uint8_t u[64];
    shake256_squeeze(&c, u, 64);
    ristretto_from_uniform_bytes(&out[i], u);
  }
}
