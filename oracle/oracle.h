/* oracle.h -- CPU restatement of the ZkVM verification hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load liboracle.so.  The product
 * (zkvm_amd/, libzkgpu.so) must never link, load or call anything here.
 *
 * PARITY UNPINNED vs interstellar/zkvm: /root/reference holds no source code
 * (a "repository moved" README and a licence), and the
 * arithmetic lives in third-party crates (curve25519-dalek, bulletproofs,
 * merlin) whose pinned versions are unknowable without a Cargo.lock.  Every
 * function therefore cites the public specification it restates, and the
 * results are pinned by:
 *   - libsodium 1.0.18 (independent ristretto255) -> tests/golden/ristretto255.json
 *   - RFC 9496 sec 4.1 decimal constants (asserted in oracle/pyref.py)
 *   - FIPS 202 via hashlib, the public Merlin "test protocol" answer,
 *     dalek's published Pedersen blinding base encoding
 *   - oracle/pyref.py (Python big-int, exact by construction)
 */
#ifndef ZK_ORACLE_H
#define ZK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- GF(2^255-19), five 51-bit limbs (RFC 9496 sec 4.1) ---------------- */
typedef struct { uint64_t v[5]; } fe;

void fe_frombytes(fe *h, const uint8_t s[32]);      /* ignores bit 255 */
void fe_tobytes(uint8_t s[32], const fe *h);        /* canonical */
void fe_add(fe *h, const fe *f, const fe *g);
void fe_sub(fe *h, const fe *f, const fe *g);
void fe_neg(fe *h, const fe *f);
void fe_mul(fe *h, const fe *f, const fe *g);
void fe_sq(fe *h, const fe *f);
void fe_invert(fe *h, const fe *f);
void fe_pow22523(fe *h, const fe *f);               /* f^((p-5)/8) */
int fe_is_negative(const fe *f);
int fe_is_zero(const fe *f);
int fe_eq(const fe *f, const fe *g);
int fe_sqrt_ratio_m1(fe *r, const fe *u, const fe *v); /* RFC 9496 sec 4.2 */

/* ---- scalars mod l (RFC 9496 sec 4.4), 4 x u64 little endian, canonical -- */
typedef struct { uint64_t v[4]; } sc;

void sc_from_bytes_wide(sc *r, const uint8_t b[64]);
void sc_from_bytes_mod_order(sc *r, const uint8_t b[32]);
int sc_from_canonical_bytes(sc *r, const uint8_t b[32]); /* 1 ok, 0 reject */
void sc_to_bytes(uint8_t b[32], const sc *a);
void sc_from_u64(sc *r, uint64_t x);
void sc_add(sc *r, const sc *a, const sc *b);
void sc_sub(sc *r, const sc *a, const sc *b);
void sc_neg(sc *r, const sc *a);
void sc_mul(sc *r, const sc *a, const sc *b);
void sc_invert(sc *r, const sc *a);
int sc_is_zero(const sc *a);
int sc_eq(const sc *a, const sc *b);

/* ---- edwards25519 extended coordinates, ristretto255 encoding ----------- */
typedef struct { fe X, Y, Z, T; } ge;

void ge_identity(ge *p);
void ge_basepoint(ge *p);
void ge_add(ge *r, const ge *p, const ge *q);
void ge_sub(ge *r, const ge *p, const ge *q);
void ge_double(ge *r, const ge *p);
void ge_neg(ge *r, const ge *p);
void ge_scalarmult(ge *r, const sc *k, const ge *p);
int ge_ristretto_eq(const ge *p, const ge *q);
int ge_is_identity(const ge *p);
int ristretto_decode(ge *p, const uint8_t s[32]);   /* 1 ok, 0 reject; sec 4.3.1 */
void ristretto_encode(uint8_t s[32], const ge *p);  /* sec 4.3.2 */
void ristretto_from_uniform_bytes(ge *p, const uint8_t b[64]); /* sec 4.3.4 */

/* ---- multiscalar multiplication (restates curve25519-dalek's
 *      VartimeMultiscalarMul: Straus below 190 terms, Pippenger above) ---- */
void ge_msm_vartime(ge *r, const sc *scalars, const ge *points, size_t n);
void ge_msm_straus(ge *r, const sc *scalars, const ge *points, size_t n);
void ge_msm_pippenger(ge *r, const sc *scalars, const ge *points, size_t n);
void ge_msm_naive(ge *r, const sc *scalars, const ge *points, size_t n);

/* ---- byte-level entry points mirroring the zkgpu C ABI (include/zkgpu.h) --
 * Return 0 on success, ZKO_EINVALID_POINT (+ index via bad_index) when a
 * point fails to decode. */
#define ZKO_OK 0
#define ZKO_EINVALID_POINT (-2)
int zko_msm(const uint8_t *scalars, const uint8_t *points, size_t n,
            uint8_t out[32], size_t *bad_index);
/* accept bit i = 1 iff every point of MSM i decodes and the MSM is the
 * identity.  threads <= 1: scalar; > 1: OpenMP over independent MSMs. */
int zko_verify_batch(const uint8_t *scalars, const uint8_t *points,
                     const uint64_t *offsets, size_t batch,
                     uint8_t *accept_bitmap, int threads);
int zko_decode_batch(const uint8_t *points, size_t n, uint8_t *ok);
int zko_max_threads(void);

/* ---- FIPS 202 ------------------------------------------------------------ */
void keccak_f1600(uint64_t st[25]);
void sha3_512(uint8_t out[64], const uint8_t *in, size_t len);
typedef struct { uint64_t st[25]; unsigned pos; int squeezing; } shake256_ctx;
void shake256_init(shake256_ctx *c);
void shake256_absorb(shake256_ctx *c, const uint8_t *in, size_t len);
void shake256_squeeze(shake256_ctx *c, uint8_t *out, size_t len);

/* ---- STROBE-128 / Merlin v1.0 (merlin.cool) ------------------------------- */
typedef struct {
  uint8_t st[200];
  uint8_t pos, pos_begin, cur_flags;
} strobe128;
typedef struct { strobe128 s; } merlin_transcript;

void merlin_init(merlin_transcript *t, const uint8_t *label, size_t len);
void merlin_append_message(merlin_transcript *t, const char *label,
                           const uint8_t *msg, size_t len);
void merlin_append_u64(merlin_transcript *t, const char *label, uint64_t x);
void merlin_challenge_bytes(merlin_transcript *t, const char *label,
                            uint8_t *out, size_t len);
void merlin_append_scalar(merlin_transcript *t, const char *label, const sc *s);
void merlin_append_point(merlin_transcript *t, const char *label, const uint8_t p[32]);
void merlin_challenge_scalar(merlin_transcript *t, const char *label, sc *out);
/* witness-rekeyed RNG used by provers (merlin "TranscriptRng") */
void merlin_rekey_with_witness(merlin_transcript *t, const char *label,
                               const uint8_t *w, size_t len);
void merlin_finalize_rng(merlin_transcript *t, const uint8_t rng_seed[32]);
void merlin_rng_fill(merlin_transcript *t, uint8_t *out, size_t len);

/* ---- Bulletproofs generators ---------------------------------------------- */
void pedersen_gens(ge *B, ge *B_blinding);
void bulletproof_gens_chain(ge *out, size_t n, char which /* 'G' or 'H' */, uint32_t party);

#ifdef __cplusplus
}
#endif
#endif
