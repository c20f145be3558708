// merlin_x8.hpp -- EIGHT Merlin transcripts in lockstep on the host (AVX-512): Keccak-f[1600] on eight states at once,
// one per 64-bit lane of a __m512i, under a STROBE-128 framing that is shared by the eight (same labels, same lengths,
// same positions -- only the message bytes differ).
//
// Why: the host half of Tx::verify on serialized transactions (zkvm_tx.hpp: contract ids, the transaction-ID Merkle tree,
// MuSig coefficients, the signature challenge) is ~35 Keccak-f per payment and NOTHING else of weight: 7 us per
// transaction on one EPYC core, the stage that bounds zkgpu_tx_verify_batch.  Transactions of
// one shape hash messages of the same lengths in the same order, so eight of them can share every instruction:
// vprolq rotates, vpternlogq does theta's three-way XOR and chi in one instruction each -- ~270 instructions per state
// and permutation instead of ~3600.
//
// The STROBE state of lane l is st_[w] lane l; bytes absorbed since the last permutation are collected per lane in plain
// byte buffers (XORed into the state, eight lanes at a time, when the permutation runs): absorbing is a memcpy.
// (Same algorithm as merlin.hpp / FIPS 202 / STROBE v1.0.2; tests compare the two byte for byte.)
#pragma once
#include "merlin.hpp"

#include <algorithm>

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define ZK_HAVE_X8 1
#define ZK_X8 __attribute__((target("avx512f,avx512vl")))

namespace zk {

inline bool x8_available() {
  static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
  return ok;
}

ZK_X8 inline void keccak_f1600_x8(__m512i a[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
      0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
      0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
      0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
      0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
#define ZK_X3(x, y, z) _mm512_ternarylogic_epi64((x), (y), (z), 0x96)      /* x ^ y ^ z */
#define ZK_CHI(x, y, z) _mm512_ternarylogic_epi64((x), (y), (z), 0xD2)     /* x ^ (~y & z) */
  for (int r = 0; r < 24; ++r) {
    // theta
    const __m512i c0 = ZK_X3(ZK_X3(a[0], a[5], a[10]), a[15], a[20]);
    const __m512i c1 = ZK_X3(ZK_X3(a[1], a[6], a[11]), a[16], a[21]);
    const __m512i c2 = ZK_X3(ZK_X3(a[2], a[7], a[12]), a[17], a[22]);
    const __m512i c3 = ZK_X3(ZK_X3(a[3], a[8], a[13]), a[18], a[23]);
    const __m512i c4 = ZK_X3(ZK_X3(a[4], a[9], a[14]), a[19], a[24]);
    const __m512i r0 = _mm512_rol_epi64(c0, 1), r1 = _mm512_rol_epi64(c1, 1), r2 = _mm512_rol_epi64(c2, 1),
                  r3 = _mm512_rol_epi64(c3, 1), r4 = _mm512_rol_epi64(c4, 1);
    // a[x + 5y] ^= C[x-1] ^ rol(C[x+1], 1); rho + pi: B[y][2x + 3y] = rol(.., rho[x][y])
    const __m512i b0 = ZK_X3(a[0], c4, r1);
    const __m512i b10 = _mm512_rol_epi64(ZK_X3(a[1], c0, r2), 1);
    const __m512i b20 = _mm512_rol_epi64(ZK_X3(a[2], c1, r3), 62);
    const __m512i b5 = _mm512_rol_epi64(ZK_X3(a[3], c2, r4), 28);
    const __m512i b15 = _mm512_rol_epi64(ZK_X3(a[4], c3, r0), 27);
    const __m512i b16 = _mm512_rol_epi64(ZK_X3(a[5], c4, r1), 36);
    const __m512i b1 = _mm512_rol_epi64(ZK_X3(a[6], c0, r2), 44);
    const __m512i b11 = _mm512_rol_epi64(ZK_X3(a[7], c1, r3), 6);
    const __m512i b21 = _mm512_rol_epi64(ZK_X3(a[8], c2, r4), 55);
    const __m512i b6 = _mm512_rol_epi64(ZK_X3(a[9], c3, r0), 20);
    const __m512i b7 = _mm512_rol_epi64(ZK_X3(a[10], c4, r1), 3);
    const __m512i b17 = _mm512_rol_epi64(ZK_X3(a[11], c0, r2), 10);
    const __m512i b2 = _mm512_rol_epi64(ZK_X3(a[12], c1, r3), 43);
    const __m512i b12 = _mm512_rol_epi64(ZK_X3(a[13], c2, r4), 25);
    const __m512i b22 = _mm512_rol_epi64(ZK_X3(a[14], c3, r0), 39);
    const __m512i b23 = _mm512_rol_epi64(ZK_X3(a[15], c4, r1), 41);
    const __m512i b8 = _mm512_rol_epi64(ZK_X3(a[16], c0, r2), 45);
    const __m512i b18 = _mm512_rol_epi64(ZK_X3(a[17], c1, r3), 15);
    const __m512i b3 = _mm512_rol_epi64(ZK_X3(a[18], c2, r4), 21);
    const __m512i b13 = _mm512_rol_epi64(ZK_X3(a[19], c3, r0), 8);
    const __m512i b14 = _mm512_rol_epi64(ZK_X3(a[20], c4, r1), 18);
    const __m512i b24 = _mm512_rol_epi64(ZK_X3(a[21], c0, r2), 2);
    const __m512i b9 = _mm512_rol_epi64(ZK_X3(a[22], c1, r3), 61);
    const __m512i b19 = _mm512_rol_epi64(ZK_X3(a[23], c2, r4), 56);
    const __m512i b4 = _mm512_rol_epi64(ZK_X3(a[24], c3, r0), 14);
    // chi, iota
    a[0] = _mm512_xor_si512(ZK_CHI(b0, b1, b2), _mm512_set1_epi64((long long)RC[r]));
    a[1] = ZK_CHI(b1, b2, b3); a[2] = ZK_CHI(b2, b3, b4); a[3] = ZK_CHI(b3, b4, b0); a[4] = ZK_CHI(b4, b0, b1);
    a[5] = ZK_CHI(b5, b6, b7); a[6] = ZK_CHI(b6, b7, b8); a[7] = ZK_CHI(b7, b8, b9); a[8] = ZK_CHI(b8, b9, b5); a[9] = ZK_CHI(b9, b5, b6);
    a[10] = ZK_CHI(b10, b11, b12); a[11] = ZK_CHI(b11, b12, b13); a[12] = ZK_CHI(b12, b13, b14); a[13] = ZK_CHI(b13, b14, b10); a[14] = ZK_CHI(b14, b10, b11);
    a[15] = ZK_CHI(b15, b16, b17); a[16] = ZK_CHI(b16, b17, b18); a[17] = ZK_CHI(b17, b18, b19); a[18] = ZK_CHI(b18, b19, b15); a[19] = ZK_CHI(b19, b15, b16);
    a[20] = ZK_CHI(b20, b21, b22); a[21] = ZK_CHI(b21, b22, b23); a[22] = ZK_CHI(b22, b23, b24); a[23] = ZK_CHI(b23, b24, b20); a[24] = ZK_CHI(b24, b20, b21);
  }
#undef ZK_X3
#undef ZK_CHI
}

// Eight transcripts that were ONE transcript up to now (a copy of `proto`) and from here on receive messages of equal
// lengths with contents of their own.
class TranscriptX8 {
 public:
  ZK_X8 explicit TranscriptX8(const Transcript& proto) {
    uint32_t w[52];
    proto.export_state(w);
    for (int i = 0; i < 25; ++i) st_[i] = _mm512_set1_epi64((long long)((uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32)));
    pos_ = w[50]; pos_begin_ = w[51];
    std::memset(buf_, 0, sizeof buf_);
  }
  // one message (label and total length the same for the eight), given in pieces: begin, any number of data calls, no end call
  ZK_X8 void begin_message(const char* label, size_t total_len) {
    const uint8_t len[4] = {(uint8_t)total_len, (uint8_t)(total_len >> 8), (uint8_t)(total_len >> 16), (uint8_t)(total_len >> 24)};
    begin_op(kM | kA);
    same((const uint8_t*)label, std::strlen(label));
    same(len, 4);
    begin_op(kA);
  }
  ZK_X8 void data(const uint8_t* const d[8], size_t n) {
    size_t at = 0;
    while (at < n) {
      const size_t take = std::min<size_t>(n - at, kRate - pos_);
      for (int l = 0; l < 8; ++l) xor_bytes(buf_[l] + pos_, d[l] + at, take);
      pos_ += (unsigned)take; at += take;
      if (pos_ == kRate) run_f();
    }
  }
  ZK_X8 void data_same(const uint8_t* d, size_t n) { same(d, n); }
  // n <= 64 challenge bytes per lane
  ZK_X8 void challenge(const char* label, uint8_t* const out[8], size_t n) {
    const uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    begin_op(kM | kA);
    same((const uint8_t*)label, std::strlen(label));
    same(len, 4);
    begin_op(kI | kA | kC);               // always permutes: two framing bytes were just absorbed
    alignas(64) uint64_t w[8];
    for (size_t q = 0; q < (n + 7) / 8; ++q) {
      _mm512_store_si512((void*)w, st_[q]);
      const size_t take = std::min<size_t>(8, n - 8 * q);
      for (int l = 0; l < 8; ++l) std::memcpy(out[l] + 8 * q, &w[l], take);
      // the squeezed bytes are zeroed (STROBE's PRF): whole words when n is a multiple of 8, else the low bytes
      st_[q] = take == 8 ? _mm512_setzero_si512() : _mm512_and_si512(st_[q], _mm512_set1_epi64((long long)(~0ULL << (8 * take))));
    }
    pos_ = (unsigned)n;
  }

 private:
  static constexpr unsigned kRate = 166;
  static constexpr uint8_t kI = 1, kA = 2, kC = 4, kM = 16, kK = 32;
  __m512i st_[25];
  alignas(64) uint8_t buf_[8][192];         // bytes to XOR into the rate (168 used; three 64-byte blocks), per lane
  unsigned pos_ = 0, pos_begin_ = 0;

  // the buffers are zero wherever nothing has been absorbed since the last permutation, and every position is written at
  // most once before the next: XORing the bytes in is copying them (32 and 8 bytes at a time are what the plans mostly hold)
  ZK_X8 static void xor_bytes(uint8_t* dst, const uint8_t* src, size_t n) {
    if (n == 32) { _mm256_storeu_si256((__m256i*)dst, _mm256_loadu_si256((const __m256i*)src)); return; }
    if (n == 8) { uint64_t w; std::memcpy(&w, src, 8); std::memcpy(dst, &w, 8); return; }
    std::memcpy(dst, src, n);
  }
  ZK_X8 void same(const uint8_t* d, size_t n) {
    size_t at = 0;
    while (at < n) {
      const size_t take = std::min<size_t>(n - at, kRate - pos_);
      for (int l = 0; l < 8; ++l) xor_bytes(buf_[l] + pos_, d + at, take);
      pos_ += (unsigned)take; at += take;
      if (pos_ == kRate) run_f();
    }
  }
  ZK_X8 void run_f() {
    for (int l = 0; l < 8; ++l) { buf_[l][pos_] ^= (uint8_t)pos_begin_; buf_[l][pos_ + 1] ^= 0x04; buf_[l][kRate + 1] ^= 0x80; }
    // lane l's 64-byte block j holds its words 8 j .. 8 j + 7; state word w wants word w of all eight lanes: an 8 x 8
    // transpose of 64-bit elements per block (8 unpacks, 8 two-source permutes, 8 lane shuffles -- a gather per word
    // costs three times that)
    const __m512i IA = _mm512_setr_epi64(0, 1, 8, 9, 2, 3, 10, 11), IB = _mm512_setr_epi64(4, 5, 12, 13, 6, 7, 14, 15);
    for (int j = 0; j < 3; ++j) {
      __m512i r[8], a[4], b[4];
      for (int l = 0; l < 8; ++l) r[l] = _mm512_load_si512((const void*)(buf_[l] + 64 * j));
      for (int k = 0; k < 4; ++k) { a[k] = _mm512_unpacklo_epi64(r[2 * k], r[2 * k + 1]); b[k] = _mm512_unpackhi_epi64(r[2 * k], r[2 * k + 1]); }
      const __m512i pa_lo = _mm512_permutex2var_epi64(a[0], IA, a[1]), pa_hi = _mm512_permutex2var_epi64(a[0], IB, a[1]);
      const __m512i qa_lo = _mm512_permutex2var_epi64(a[2], IA, a[3]), qa_hi = _mm512_permutex2var_epi64(a[2], IB, a[3]);
      const __m512i pb_lo = _mm512_permutex2var_epi64(b[0], IA, b[1]), pb_hi = _mm512_permutex2var_epi64(b[0], IB, b[1]);
      const __m512i qb_lo = _mm512_permutex2var_epi64(b[2], IA, b[3]), qb_hi = _mm512_permutex2var_epi64(b[2], IB, b[3]);
      __m512i col[8];
      col[0] = _mm512_shuffle_i64x2(pa_lo, qa_lo, 0x44); col[2] = _mm512_shuffle_i64x2(pa_lo, qa_lo, 0xEE);
      col[4] = _mm512_shuffle_i64x2(pa_hi, qa_hi, 0x44); col[6] = _mm512_shuffle_i64x2(pa_hi, qa_hi, 0xEE);
      col[1] = _mm512_shuffle_i64x2(pb_lo, qb_lo, 0x44); col[3] = _mm512_shuffle_i64x2(pb_lo, qb_lo, 0xEE);
      col[5] = _mm512_shuffle_i64x2(pb_hi, qb_hi, 0x44); col[7] = _mm512_shuffle_i64x2(pb_hi, qb_hi, 0xEE);
      const int n = j < 2 ? 8 : 5;                        // words 16 .. 20 of the rate in the last block
      for (int c = 0; c < n; ++c) st_[8 * j + c] = _mm512_xor_si512(st_[8 * j + c], col[c]);
    }
    std::memset(buf_, 0, sizeof buf_);
    keccak_f1600_x8(st_);
    pos_ = 0;
    pos_begin_ = 0;
  }
  ZK_X8 void begin_op(uint8_t flags) {
    const uint8_t hdr[2] = {(uint8_t)pos_begin_, flags};
    pos_begin_ = pos_ + 1;
    same(hdr, 2);
    if ((flags & (kC | kK)) && pos_ != 0) run_f();
  }
};

}  // namespace zk
#else
#define ZK_HAVE_X8 0
namespace zk { inline bool x8_available() { return false; } }
#endif
