/* keccak.c -- Keccak-f[1600], SHA3-512, SHAKE256 (FIPS 202).
 *
 * TEST INFRASTRUCTURE (see oracle.h).  The reference path reaches these through
 * the `sha3`/`keccak` crates (not mounted).
 * Pinned against Python's hashlib in tests/test_oracle_transcript.py.
 */
#include "oracle.h"
#include <string.h>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
    0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};

static inline uint64_t rol(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

void keccak_f1600(uint64_t a[25]) {
  /* rho offsets indexed [x + 5y] */
  static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39,
                              41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  for (int rnd = 0; rnd < 24; ++rnd) {
    uint64_t c[5], d[5], b[25];
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
    for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
    for (int x = 0; x < 5; ++x)
      for (int y = 0; y < 5; ++y) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], RHO[x + 5 * y]);
    for (int y = 0; y < 5; ++y)
      for (int x = 0; x < 5; ++x) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
    a[0] ^= RC[rnd];
  }
}

static void xor_byte(uint64_t st[25], unsigned pos, uint8_t b) { st[pos / 8] ^= (uint64_t)b << (8 * (pos % 8)); }
static uint8_t get_byte(const uint64_t st[25], unsigned pos) { return (uint8_t)(st[pos / 8] >> (8 * (pos % 8))); }

void sha3_512(uint8_t out[64], const uint8_t *in, size_t len) {
  uint64_t st[25];
  const unsigned rate = 72;
  unsigned pos = 0;
  memset(st, 0, sizeof st);
  for (size_t i = 0; i < len; ++i) {
    xor_byte(st, pos++, in[i]);
    if (pos == rate) { keccak_f1600(st); pos = 0; }
  }
  xor_byte(st, pos, 0x06);
  xor_byte(st, rate - 1, 0x80);
  keccak_f1600(st);
  for (unsigned i = 0; i < 64; ++i) out[i] = get_byte(st, i);
}

#define SHAKE256_RATE 136
void shake256_init(shake256_ctx *c) { memset(c, 0, sizeof *c); }

void shake256_absorb(shake256_ctx *c, const uint8_t *in, size_t len) {
  for (size_t i = 0; i < len; ++i) {
    xor_byte(c->st, c->pos++, in[i]);
    if (c->pos == SHAKE256_RATE) { keccak_f1600(c->st); c->pos = 0; }
  }
}

void shake256_squeeze(shake256_ctx *c, uint8_t *out, size_t len) {
  if (!c->squeezing) {
    xor_byte(c->st, c->pos, 0x1F);
    xor_byte(c->st, SHAKE256_RATE - 1, 0x80);
    keccak_f1600(c->st);
    c->pos = 0;
    c->squeezing = 1;
  }
  for (size_t i = 0; i < len; ++i) {
    if (c->pos == SHAKE256_RATE) { keccak_f1600(c->st); c->pos = 0; }
    out[i] = get_byte(c->st, c->pos++);
  }
}
