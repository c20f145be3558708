/* msm.c -- variable-time multiscalar multiplication on the CPU.
 *
 * TEST INFRASTRUCTURE (see oracle.h).  Restates the algorithm family of
 * curve25519-dalek's `VartimeMultiscalarMul for RistrettoPoint`
 * (source not mounted): width-5 NAF Straus for
 * fewer than 190 terms, signed-digit radix-2^w Pippenger (w = 6, 7, 8 by size)
 * above -- Pippenger 1976; Bernstein-Doumen-Lange-Oosterwijk 2012.  The result
 * of an MSM is a group element, so parity is on the canonical encoding, not on
 * the schedule; this file is also what bench.py times as `cpu_baseline`
 * (kind "port").
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

void ge_msm_naive(ge *r, const sc *scalars, const ge *points, size_t n) {
  ge acc, t;
  ge_identity(&acc);
  for (size_t i = 0; i < n; ++i) {
    ge_scalarmult(&t, &scalars[i], &points[i]);
    ge_add(&acc, &acc, &t);
  }
  *r = acc;
}

/* width-w non-adjacent form: digits odd, |d| < 2^(w-1), at most one non-zero
 * in any w consecutive positions.  naf has 257 entries. */
static void sc_naf(int8_t naf[257], const sc *k, int w) {
  uint64_t x[5] = {k->v[0], k->v[1], k->v[2], k->v[3], 0};
  memset(naf, 0, 257);
  int width = 1 << w;
  int pos = 0;
  uint64_t carry = 0;
  while (pos < 257) {
    int idx = pos / 64, bit = pos % 64;
    uint64_t buf = x[idx] >> bit;
    if (bit > 64 - w && idx < 4) buf |= x[idx + 1] << (64 - bit);
    uint64_t window = carry + (buf & (uint64_t)(width - 1));
    if ((window & 1) == 0) { pos += 1; continue; }
    if (window < (uint64_t)(width / 2)) { carry = 0; naf[pos] = (int8_t)window; }
    else { carry = 1; naf[pos] = (int8_t)((int)window - width); }
    pos += w;
  }
}

void ge_msm_straus(ge *r, const sc *scalars, const ge *points, size_t n) {
  /* tables of odd multiples P, 3P, ..., 15P */
  ge(*tab)[8] = malloc(sizeof(ge[8]) * (n ? n : 1));
  int8_t(*naf)[257] = malloc(257 * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) {
    ge p2;
    tab[i][0] = points[i];
    ge_double(&p2, &points[i]);
    for (int j = 1; j < 8; ++j) ge_add(&tab[i][j], &tab[i][j - 1], &p2);
    sc_naf(naf[i], &scalars[i], 5);
  }
  ge acc;
  ge_identity(&acc);
  int top = 256;
  for (; top >= 0; --top) {
    int any = 0;
    for (size_t i = 0; i < n && !any; ++i) any |= naf[i][top] != 0;
    if (any) break;
  }
  for (int b = top; b >= 0; --b) {
    ge_double(&acc, &acc);
    for (size_t i = 0; i < n; ++i) {
      int d = naf[i][b];
      if (d > 0) ge_add(&acc, &acc, &tab[i][d / 2]);
      else if (d < 0) ge_sub(&acc, &acc, &tab[i][(-d) / 2]);
    }
  }
  *r = acc;
  free(tab);
  free(naf);
}

/* signed radix-2^w digits, d_j in [-2^(w-1), 2^(w-1)), sum d_j 2^(wj) = k */
static int sc_radix_2w(int16_t *digits, const sc *k, int w) {
  int nd = (256 + w - 1) / w + 1;
  uint64_t x[5] = {k->v[0], k->v[1], k->v[2], k->v[3], 0};
  int carry = 0;
  for (int j = 0; j < nd; ++j) {
    int pos = j * w;
    int v = carry;
    if (pos < 256) {
      int idx = pos / 64, bit = pos % 64;
      uint64_t buf = x[idx] >> bit;
      if (bit > 64 - w) buf |= x[idx + 1] << (64 - bit);
      v += (int)(buf & ((1u << w) - 1));
    }
    carry = (v + (1 << (w - 1))) >> w;
    digits[j] = (int16_t)(v - (carry << w));
  }
  return nd;
}

void ge_msm_pippenger(ge *r, const sc *scalars, const ge *points, size_t n) {
  int w = n < 500 ? 6 : (n < 800 ? 7 : 8);
  int nb = 1 << (w - 1);
  int nd = (256 + w - 1) / w + 1;
  int16_t *digits = malloc(sizeof(int16_t) * (size_t)nd * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) sc_radix_2w(digits + i * (size_t)nd, &scalars[i], w);
  ge *buckets = malloc(sizeof(ge) * (size_t)nb);
  ge total;
  ge_identity(&total);
  for (int j = nd - 1; j >= 0; --j) {
    for (int b = 0; b < nb; ++b) ge_identity(&buckets[b]);
    for (size_t i = 0; i < n; ++i) {
      int d = digits[i * (size_t)nd + j];
      if (d > 0) ge_add(&buckets[d - 1], &buckets[d - 1], &points[i]);
      else if (d < 0) ge_sub(&buckets[-d - 1], &buckets[-d - 1], &points[i]);
    }
    /* sum_b (b+1) * bucket[b] by running sums */
    ge run = buckets[nb - 1], col = buckets[nb - 1];
    for (int b = nb - 2; b >= 0; --b) {
      ge_add(&run, &run, &buckets[b]);
      ge_add(&col, &col, &run);
    }
    for (int s = 0; s < w; ++s) ge_double(&total, &total);
    ge_add(&total, &total, &col);
  }
  *r = total;
  free(buckets);
  free(digits);
}

void ge_msm_vartime(ge *r, const sc *scalars, const ge *points, size_t n) {
  if (n < 190) ge_msm_straus(r, scalars, points, n);
  else ge_msm_pippenger(r, scalars, points, n);
}

/* ---- byte-level entry points (same shapes as include/zkgpu.h) ------------ */

int zko_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

int zko_decode_batch(const uint8_t *points, size_t n, uint8_t *ok) {
  ge p;
  for (size_t i = 0; i < n; ++i) ok[i] = (uint8_t)ristretto_decode(&p, points + 32 * i);
  return ZKO_OK;
}

static int msm_bytes(ge *out, const uint8_t *scalars, const uint8_t *points, size_t n, size_t *bad) {
  sc *s = malloc(sizeof(sc) * (n ? n : 1));
  ge *p = malloc(sizeof(ge) * (n ? n : 1));
  int rc = ZKO_OK;
  for (size_t i = 0; i < n; ++i) {
    sc_from_bytes_mod_order(&s[i], scalars + 32 * i);
    if (!ristretto_decode(&p[i], points + 32 * i)) {
      if (bad) *bad = i;
      rc = ZKO_EINVALID_POINT;
      break;
    }
  }
  if (rc == ZKO_OK) ge_msm_vartime(out, s, p, n);
  free(s);
  free(p);
  return rc;
}

int zko_msm(const uint8_t *scalars, const uint8_t *points, size_t n, uint8_t out[32], size_t *bad_index) {
  ge r;
  int rc = msm_bytes(&r, scalars, points, n, bad_index);
  if (rc != ZKO_OK) { memset(out, 0, 32); return rc; }
  ristretto_encode(out, &r);
  return ZKO_OK;
}

int zko_verify_batch(const uint8_t *scalars, const uint8_t *points, const uint64_t *offsets, size_t batch,
                     uint8_t *accept_bitmap, int threads) {
  uint8_t *acc = calloc(batch ? batch : 1, 1);
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 1 ? threads : 1)
#endif
  for (long long i = 0; i < (long long)batch; ++i) {
    ge r;
    size_t o = offsets[i], n = offsets[i + 1] - offsets[i];
    int rc = msm_bytes(&r, scalars + 32 * o, points + 32 * o, n, NULL);
    acc[i] = (uint8_t)(rc == ZKO_OK && ge_is_identity(&r));
  }
  memset(accept_bitmap, 0, (batch + 7) / 8);
  for (size_t i = 0; i < batch; ++i)
    if (acc[i]) accept_bitmap[i / 8] |= (uint8_t)(1u << (i % 8));
  free(acc);
  return ZKO_OK;
}
