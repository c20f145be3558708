/* zkvm_tx.c -- ZkVM transactions of the PAYMENT SUBSET: build, identify, verify.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY UNPINNED: no source and no specification is mounted under
 * /root/reference; the wire format, opcodes, labels and hashing restate a recollection of the public ZkVM design notes
 * (DESIGN.md sec 4.5 describes the subset), written here independently of zkvm_amd/csrc/zkvm_tx.hpp so that the two
 * check each other.
 *
 *   Tx       := version:u64 | mintime_ms:u64 | maxtime_ms:u64 | n:u32 program[n] | R:32 s:32 | n:u32 proof[n]
 *   Contract := anchor:32 | predicate:32 | k:u32 | item*     item := 0x00 n:u32 bytes[n] | 0x02 qty:32 flavor:32
 *   program  :  0x00 push:n:x  0x02 drop  0x03 dup:k  0x04 roll:k  0x06 var  0x18 cloak:m:n  0x1b input  0x1c output:k
 *               0x20 signtx
 * Transaction ID: Merkle root (RFC 6962 shape, Merlin transcripts as the hash) over the log
 *   [header, input ids and output ids in program order]; signature: Schnorr over ristretto255 with the MuSig
 *   aggregation of all signtx keys, message = transaction ID. */
#include "oracle.h"
#include "r1cs.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

int zko_cloak_prove(const uint64_t *q, const uint8_t *flavors, size_t n_in, size_t n_out, const uint8_t seed[32],
                    uint8_t *commitments, uint8_t *proof, size_t proof_cap, size_t *proof_len, size_t *n_multipliers);
int zko_cloak_verify(const uint8_t *commitments, size_t n_in, size_t n_out, const uint8_t *proof, size_t proof_len,
                     const uint8_t r_bytes[64]);

/* No capacity of this restatement is a rule of the format: the stack, the log and the key list grow with the program
 * (until round 3 they were fixed arrays -- 256 items, 160 log entries, 64 keys -- and a longer transaction was reported
 * INVALID, which the product's VM, having no such limits, did not do: ADVICE r03). */

enum { IT_DATA = 0, IT_VAR = 1, IT_VALUE = 2, IT_CONTRACT = 3 };
typedef struct { int kind; const uint8_t *p; size_t n; uint8_t own[64]; } item;   /* a value lives in own[] (p unused), everything else points into the transaction */
typedef struct { int kind; uint64_t a, b, c; uint8_t id[32]; } entry;              /* 0 header, 1 input, 2 output */

static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
static void put32(uint8_t *p, uint32_t v) { for (int i = 0; i < 4; ++i) p[i] = (uint8_t)(v >> (8 * i)); }
static void put64(uint8_t *p, uint64_t v) { for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(v >> (8 * i)); }

static void cid(const uint8_t *ser, size_t n, uint8_t out[32]) {
  merlin_transcript t;
  merlin_init(&t, (const uint8_t *)"ZkVM.contractid", 15);
  merlin_append_message(&t, "contract", ser, n);
  merlin_challenge_bytes(&t, "id", out, 32);
}
static void ratchet(const uint8_t old_anchor[32], uint8_t out[32]) {
  merlin_transcript t;
  merlin_init(&t, (const uint8_t *)"ZkVM.ratchet-anchor", 19);
  merlin_append_message(&t, "old", old_anchor, 32);
  merlin_challenge_bytes(&t, "new", out, 32);
}

static void leaf_hash(const entry *e, uint8_t out[32]) {
  merlin_transcript t;
  merlin_init(&t, (const uint8_t *)"ZkVM.txid", 9);
  if (e->kind == 0) {
    merlin_append_u64(&t, "tx.version", e->a);
    merlin_append_u64(&t, "tx.mintime", e->b);
    merlin_append_u64(&t, "tx.maxtime", e->c);
  } else {
    merlin_append_message(&t, e->kind == 1 ? "input" : "output", e->id, 32);
  }
  merlin_challenge_bytes(&t, "merkle.leaf", out, 32);
}
/* root of entries [lo, hi): the left subtree takes the largest power of two below the count */
static void root_of(const entry *e, size_t lo, size_t hi, uint8_t out[32]) {
  size_t n = hi - lo;
  if (n == 0) {
    merlin_transcript t;
    merlin_init(&t, (const uint8_t *)"ZkVM.txid", 9);
    merlin_challenge_bytes(&t, "merkle.empty", out, 32);
    return;
  }
  if (n == 1) { leaf_hash(&e[lo], out); return; }
  size_t k = 1;
  while ((k << 1) < n) k <<= 1;
  uint8_t l[32], r[32];
  root_of(e, lo, lo + k, l);
  root_of(e, lo + k, hi, r);
  merlin_transcript t;
  merlin_init(&t, (const uint8_t *)"ZkVM.txid", 9);
  merlin_append_message(&t, "L", l, 32);
  merlin_append_message(&t, "R", r, 32);
  merlin_challenge_bytes(&t, "merkle.node", out, 32);
}

/* walks a serialized contract; returns the number of payload items or -1 (malformed) / -2 (outside the subset);
 * items (optional) receive the payload in order */
static long walk_contract(const uint8_t *p, size_t n, item *items, long cap) {
  if (n < 68) return -1;
  uint32_t k = le32(p + 64);
  size_t pos = 68;
  long count = 0;
  for (uint32_t i = 0; i < k; ++i) {
    if (pos >= n) return -1;
    uint8_t type = p[pos++];
    if (type == 0x00) {
      if (n - pos < 4) return -1;
      uint32_t len = le32(p + pos);
      pos += 4;
      if (n - pos < len) return -1;
      if (items && count < cap) { items[count].kind = IT_DATA; items[count].p = p + pos; items[count].n = len; }
      pos += len;
    } else if (type == 0x02) {
      if (n - pos < 64) return -1;
      if (items && count < cap) { items[count].kind = IT_VALUE; memcpy(items[count].own, p + pos, 64); items[count].p = NULL; items[count].n = 64; }
      pos += 64;
    } else if (type == 0x01) {
      return -2;
    } else {
      return -1;
    }
    if (++count > cap) return -1;
  }
  return pos == n ? count : -1;
}

typedef struct {
  uint64_t version, mintime, maxtime;
  uint8_t txid[32];
  size_t n_in, n_out;
  uint8_t commitments[64 * 128];
  const uint8_t *proof; size_t proof_len;
  const uint8_t *sig;
  size_t n_keys, keys_cap;
  uint8_t (*keys)[32];
} tx_run;
static void tx_run_free(tx_run *r) { if (r) { free(r->keys); free(r); } }

typedef struct { item *stack; size_t sp, stack_cap; entry *log; size_t nlog, log_cap; item *payload; } vm_mem;
static int stack_room(vm_mem *v, size_t more) {
  if (v->sp + more <= v->stack_cap) return 1;
  size_t cap = v->stack_cap ? v->stack_cap : 64;
  while (cap < v->sp + more) cap *= 2;
  item *p = realloc(v->stack, cap * sizeof(item));
  if (!p) return 0;
  v->stack = p; v->stack_cap = cap;
  return 1;
}
static int log_room(vm_mem *v) {
  if (v->nlog < v->log_cap) return 1;
  size_t cap = v->log_cap ? 2 * v->log_cap : 64;
  entry *p = realloc(v->log, cap * sizeof(entry));
  if (!p) return 0;
  v->log = p; v->log_cap = cap;
  return 1;
}

/* 0 ok, 1 invalid, 2 unsupported, 3 out of memory (never seen by the tests: reported as such, not as a verdict) */
static int run_vm(const uint8_t *tx, size_t len, tx_run *out, vm_mem *v) {
  if (len < 28) return 1;
  out->version = le64(tx); out->mintime = le64(tx + 8); out->maxtime = le64(tx + 16);
  size_t pos = 24;
  uint32_t plen = le32(tx + pos); pos += 4;
  if (len - pos < plen) return 1;
  const uint8_t *prog = tx + pos; pos += plen;
  if (len - pos < 68) return 1;
  out->sig = tx + pos; pos += 64;
  uint32_t prlen = le32(tx + pos); pos += 4;
  if (len - pos != prlen) return 1;
  out->proof = tx + pos; out->proof_len = prlen;
  if (out->version != 1) return 2;
  if (out->mintime > out->maxtime) return 1;

#define stack (v->stack)
#define log (v->log)
#define sp (v->sp)
#define nlog (v->nlog)
  int cloaked = 0, have_anchor = 0;
  uint8_t anchor[32];
  if (!log_room(v)) return 3;
  log[nlog].kind = 0; log[nlog].a = out->version; log[nlog].b = out->mintime; log[nlog].c = out->maxtime; ++nlog;
  size_t pc = 0;
  while (pc < plen) {
    uint8_t op = prog[pc++];
    uint32_t k = 0, m = 0, n = 0;
    switch (op) {
      case 0x00:
        if (plen - pc < 4) return 1;
        n = le32(prog + pc); pc += 4;
        if (plen - pc < n) return 1;
        if (!stack_room(v, 1)) return 3;
        stack[sp].kind = IT_DATA; stack[sp].p = prog + pc; stack[sp].n = n; ++sp;
        pc += n;
        break;
      case 0x02:
        if (sp == 0 || stack[sp - 1].kind == IT_VALUE || stack[sp - 1].kind == IT_CONTRACT) return 1;
        --sp;
        break;
      case 0x03:
        if (plen - pc < 4) return 1;
        k = le32(prog + pc); pc += 4;
        if (k >= sp) return 1;
        if (!stack_room(v, 1)) return 3;
        if (stack[sp - 1 - k].kind == IT_VALUE || stack[sp - 1 - k].kind == IT_CONTRACT) return 1;
        stack[sp] = stack[sp - 1 - k]; ++sp;
        break;
      case 0x04: {
        if (plen - pc < 4) return 1;
        k = le32(prog + pc); pc += 4;
        if (k >= sp) return 1;
        item t = stack[sp - 1 - k];
        for (size_t i = sp - 1 - k; i < sp - 1; ++i) stack[i] = stack[i + 1];
        stack[sp - 1] = t;
        break;
      }
      case 0x06:
        if (sp == 0 || stack[sp - 1].kind != IT_DATA || stack[sp - 1].n != 32) return 1;
        stack[sp - 1].kind = IT_VAR;
        break;
      case 0x18: {
        if (plen - pc < 8) return 1;
        m = le32(prog + pc); n = le32(prog + pc + 4); pc += 8;
        if (cloaked) return 2;
        if (m == 0 || n == 0 || m > 64 || n > 64) return 2;
        if ((size_t)sp < (size_t)m + 2 * (size_t)n) return 1;
        /* stack, bottom to top: value_0 .. value_{m-1}, q_0, f_0, .., q_{n-1}, f_{n-1} */
        size_t base = sp - (m + 2 * (size_t)n);
        for (uint32_t i = 0; i < m; ++i) {
          if (stack[base + i].kind != IT_VALUE) return 1;
          memcpy(out->commitments + 64 * i, stack[base + i].own, 64);
        }
        for (uint32_t j = 0; j < 2 * n; ++j) {
          if (stack[base + m + j].kind != IT_VAR) return 1;
          memcpy(out->commitments + 64 * m + 32 * j, stack[base + m + j].p, 32);
        }
        out->n_in = m; out->n_out = n;
        sp = base;
        if (!stack_room(v, n)) return 3;
        for (uint32_t j = 0; j < n; ++j) {
          stack[sp].kind = IT_VALUE;
          memcpy(stack[sp].own, out->commitments + 64 * (m + j), 64);
          stack[sp].p = NULL; stack[sp].n = 64;
          ++sp;
        }
        cloaked = 1;
        break;
      }
      case 0x1b: {
        if (sp == 0 || stack[sp - 1].kind != IT_DATA) return 1;
        long cnt = walk_contract(stack[sp - 1].p, stack[sp - 1].n, NULL, LONG_MAX);
        if (cnt == -2) return 2;
        if (cnt < 0) return 1;
        if (!log_room(v)) return 3;
        log[nlog].kind = 1;
        cid(stack[sp - 1].p, stack[sp - 1].n, log[nlog].id);
        memcpy(anchor, log[nlog].id, 32);
        ++nlog;
        have_anchor = 1;
        stack[sp - 1].kind = IT_CONTRACT;
        break;
      }
      case 0x1c: {
        if (plen - pc < 4) return 1;
        k = le32(prog + pc); pc += 4;
        if ((size_t)sp < (size_t)k + 1) return 1;
        if (stack[sp - 1].kind != IT_DATA || stack[sp - 1].n != 32 || !have_anchor) return 1;
        if (!log_room(v)) return 3;
        const uint8_t *pred = stack[sp - 1].p;
        --sp;
        size_t total = 68;
        for (uint32_t i = 0; i < k; ++i) {
          const item *it = &stack[sp - k + i];
          if (it->kind == IT_VALUE) total += 65;
          else if (it->kind == IT_DATA) total += 5 + it->n;
          else return 1;
        }
        uint8_t *ser = malloc(total);
        if (!ser) return 3;
        uint8_t fresh[32];
        ratchet(anchor, fresh);
        memcpy(anchor, fresh, 32);
        memcpy(ser, fresh, 32);
        memcpy(ser + 32, pred, 32);
        put32(ser + 64, k);
        size_t w = 68;
        for (uint32_t i = 0; i < k; ++i) {
          const item *it = &stack[sp - k + i];
          if (it->kind == IT_VALUE) { ser[w++] = 0x02; memcpy(ser + w, it->own, 64); w += 64; }
          else { ser[w++] = 0x00; put32(ser + w, (uint32_t)it->n); w += 4; memcpy(ser + w, it->p, it->n); w += it->n; }
        }
        log[nlog].kind = 2;
        cid(ser, total, log[nlog].id);
        ++nlog;
        free(ser);
        sp -= k;
        break;
      }
      case 0x20: {
        if (sp == 0 || stack[sp - 1].kind != IT_CONTRACT) return 1;
        const uint8_t *ser = stack[sp - 1].p;
        size_t sn = stack[sp - 1].n;
        --sp;
        long cnt = walk_contract(ser, sn, NULL, LONG_MAX);      /* count first, then read into room made for it */
        if (cnt < 0) return 1;
        free(v->payload);
        v->payload = malloc(((size_t)cnt + 1) * sizeof(item));
        if (!v->payload || !stack_room(v, (size_t)cnt)) return 3;
        walk_contract(ser, sn, v->payload, cnt);
        if (out->n_keys == out->keys_cap) {
          size_t cap = out->keys_cap ? 2 * out->keys_cap : 16;
          void *p = realloc(out->keys, cap * 32);
          if (!p) return 3;
          out->keys = p; out->keys_cap = cap;
        }
        memcpy(out->keys[out->n_keys++], ser + 32, 32);
        for (long i = 0; i < cnt; ++i) {
          stack[sp] = v->payload[i];
          ++sp;
        }
        break;
      }
      default:
        return 2;
    }
  }
  if (sp != 0) return 1;
  if (!cloaked) return 2;
  if (out->n_keys == 0) return 1;
  root_of(log, 0, nlog, out->txid);
  return 0;
#undef stack
#undef log
#undef sp
#undef nlog
}
static int run_tx(const uint8_t *tx, size_t len, tx_run *out) {
  vm_mem v;
  memset(&v, 0, sizeof v);
  memset(out, 0, sizeof *out);
  int rc = run_vm(tx, len, out, &v);
  free(v.stack); free(v.log); free(v.payload);
  return rc;
}

/* MuSig factors and the aggregated key; 0 when a key does not decode */
static int aggregate(const tx_run *r, sc *factors, ge *agg) {
  merlin_transcript t;
  merlin_init(&t, (const uint8_t *)"Musig.aggregated-key", 20);
  merlin_append_u64(&t, "n", r->n_keys);
  for (size_t i = 0; i < r->n_keys; ++i) merlin_append_point(&t, "X", r->keys[i]);
  ge_identity(agg);
  for (size_t i = 0; i < r->n_keys; ++i) {
    merlin_transcript ti = t;
    merlin_append_u64(&ti, "i", i);
    merlin_challenge_scalar(&ti, "a_i", &factors[i]);
    ge X, aX;
    if (!ristretto_decode(&X, r->keys[i])) return 0;
    ge_scalarmult(&aX, &factors[i], &X);
    ge_add(agg, agg, &aX);
  }
  return 1;
}
static void sig_challenge(const uint8_t txid[32], const uint8_t X[32], const uint8_t R[32], sc *c) {
  merlin_transcript t;
  merlin_init(&t, (const uint8_t *)"ZkVM.signtx", 11);
  merlin_append_message(&t, "txid", txid, 32);
  merlin_append_message(&t, "dom-sep", (const uint8_t *)"schnorr-signature v1", 20);
  merlin_append_point(&t, "X", X);
  merlin_append_point(&t, "R", R);
  merlin_challenge_scalar(&t, "c", c);
}

/* transaction ID and the cloak's shape; returns 0 ok / 1 invalid / 2 unsupported */
int zko_tx_id(const uint8_t *tx, size_t len, uint8_t txid[32], size_t *n_in, size_t *n_out) {
  tx_run *r = malloc(sizeof *r);
  int rc = run_tx(tx, len, r);
  if (rc == 0) { memcpy(txid, r->txid, 32); if (n_in) *n_in = r->n_in; if (n_out) *n_out = r->n_out; }
  tx_run_free(r);
  return rc;
}

/* Tx::verify: 0 accepted, 1 rejected, 2 outside the subset.  r_bytes: the proof verifier's random weight. */
int zko_tx_verify(const uint8_t *tx, size_t len, const uint8_t r_bytes[64]) {
  tx_run *r = malloc(sizeof *r);
  int rc = run_tx(tx, len, r);
  if (rc) { tx_run_free(r); return rc; }
  sc *factors = malloc((r->n_keys + 1) * sizeof(sc)), s, c;
  ge X, R, B, lhs, rhs, cX;
  int ok = aggregate(r, factors, &X) && sc_from_canonical_bytes(&s, r->sig + 32) && ristretto_decode(&R, r->sig);
  if (ok) {
    uint8_t Xenc[32];
    ristretto_encode(Xenc, &X);
    sig_challenge(r->txid, Xenc, r->sig, &c);
    ge_basepoint(&B);
    ge_scalarmult(&lhs, &s, &B);
    ge_scalarmult(&cX, &c, &X);
    ge_add(&rhs, &R, &cX);
    ok = ge_ristretto_eq(&lhs, &rhs);
  }
  if (ok) ok = zko_cloak_verify(r->commitments, r->n_in, r->n_out, r->proof, r->proof_len, r_bytes);
  free(factors);
  tx_run_free(r);
  return ok ? 0 : 1;
}

static void derive3(const uint8_t seed[32], const char *tag, uint64_t i, uint8_t *out, size_t n) {
  shake256_ctx c;
  uint8_t ib[8];
  put64(ib, i);
  shake256_init(&c);
  shake256_absorb(&c, seed, 32);
  shake256_absorb(&c, (const uint8_t *)tag, strlen(tag));
  shake256_absorb(&c, ib, 8);
  shake256_squeeze(&c, out, n);
}

/* A payment around an EXISTING cloak proof: n_in unspent contracts (one value each, a key each) -> one cloak -> n_out
 * contracts (one value each).  commitments: 64 bytes per value, inputs first, as the proof commits to them; keys,
 * anchors, recipients and the nonce are derived from the seed.  Writes the signed transaction, returns its length
 * (0: does not fit). */
size_t zko_tx_wrap_payment(size_t n_in, size_t n_out, const uint8_t *com, const uint8_t *proof, size_t proof_len,
                           const uint8_t seed[32], uint64_t mintime, uint64_t maxtime, uint8_t *out, size_t cap) {
  if (n_in == 0 || n_out == 0 || n_in > 16 || n_out > 16) return 0;
  uint8_t *prog = malloc(16384);
  size_t pl = 0;
  sc x[16];
  ge B;
  ge_basepoint(&B);
  for (size_t i = 0; i < n_in; ++i) {
    uint8_t wide[64], contract[32 + 32 + 4 + 65];
    derive3(seed, "anchor", i, contract, 32);
    derive3(seed, "key", i, wide, 64);
    sc_from_bytes_wide(&x[i], wide);
    ge P;
    ge_scalarmult(&P, &x[i], &B);
    ristretto_encode(contract + 32, &P);
    put32(contract + 64, 1);
    contract[68] = 0x02;
    memcpy(contract + 69, com + 64 * i, 64);
    prog[pl++] = 0x00; put32(prog + pl, sizeof contract); pl += 4; memcpy(prog + pl, contract, sizeof contract); pl += sizeof contract;
    prog[pl++] = 0x1b;
    prog[pl++] = 0x20;
  }
  for (size_t j = 0; j < n_out; ++j)
    for (int h = 0; h < 2; ++h) {
      prog[pl++] = 0x00; put32(prog + pl, 32); pl += 4; memcpy(prog + pl, com + 64 * (n_in + j) + 32 * h, 32); pl += 32;
      prog[pl++] = 0x06;
    }
  prog[pl++] = 0x18; put32(prog + pl, (uint32_t)n_in); pl += 4; put32(prog + pl, (uint32_t)n_out); pl += 4;
  for (size_t j = n_out; j-- > 0;) {       /* the last output value is on top */
    uint8_t wide[64], pred[32];
    sc y;
    ge P;
    derive3(seed, "recipient", j, wide, 64);
    sc_from_bytes_wide(&y, wide);
    ge_scalarmult(&P, &y, &B);
    ristretto_encode(pred, &P);
    prog[pl++] = 0x00; put32(prog + pl, 32); pl += 4; memcpy(prog + pl, pred, 32); pl += 32;
    prog[pl++] = 0x1c; put32(prog + pl, 1); pl += 4;
  }
  size_t total = 24 + 4 + pl + 64 + 4 + proof_len;
  if (total > cap) { free(prog); return 0; }
  put64(out, 1); put64(out + 8, mintime); put64(out + 16, maxtime);
  put32(out + 24, (uint32_t)pl);
  memcpy(out + 28, prog, pl);
  uint8_t *sig = out + 28 + pl;
  memset(sig, 0, 64);
  put32(sig + 64, (uint32_t)proof_len);
  memcpy(sig + 68, proof, proof_len);
  /* sign: run the program for the transaction ID and the keys, then s = r + c sum a_i x_i */
  tx_run *run = malloc(sizeof *run);
  size_t ret = 0;
  if (run_tx(out, total, run) == 0 && run->n_keys == n_in) {
    sc factors[16], xsum, nonce, c, t;       /* n_in <= 16 here */
    ge X, R;
    if (aggregate(run, factors, &X)) {
      uint8_t wide[64], Xenc[32];
      sc_from_u64(&xsum, 0);
      for (size_t i = 0; i < n_in; ++i) { sc_mul(&t, &factors[i], &x[i]); sc_add(&xsum, &xsum, &t); }
      derive3(seed, "nonce", 0, wide, 64);
      sc_from_bytes_wide(&nonce, wide);
      ge_scalarmult(&R, &nonce, &B);
      ristretto_encode(sig, &R);
      ristretto_encode(Xenc, &X);
      sig_challenge(run->txid, Xenc, sig, &c);
      sc_mul(&t, &c, &xsum);
      sc_add(&t, &t, &nonce);
      sc_to_bytes(sig + 32, &t);
      ret = total;
    }
  }
  tx_run_free(run); free(prog);
  return ret;
}

/* The same with the cloak proved here: quantities / flavors, inputs first, as zko_cloak_prove takes them; blinding
 * factors derived from the seed (0: proving failed / does not fit). */
size_t zko_tx_build_payment(size_t n_in, size_t n_out, const uint64_t *quantities, const uint8_t *flavors, const uint8_t seed[32],
                            uint64_t mintime, uint64_t maxtime, uint8_t *out, size_t cap) {
  if (n_in == 0 || n_out == 0 || n_in > 16 || n_out > 16) return 0;
  size_t nv = n_in + n_out;
  uint8_t *com = malloc(64 * nv), *proof = malloc(4096);
  size_t proof_len = 0, ret = 0;
  if (zko_cloak_prove(quantities, flavors, n_in, n_out, seed, com, proof, 4096, &proof_len, NULL) == 0)
    ret = zko_tx_wrap_payment(n_in, n_out, com, proof, proof_len, seed, mintime, maxtime, out, cap);
  free(com); free(proof);
  return ret;
}
