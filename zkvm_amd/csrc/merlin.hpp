// merlin.hpp -- STROBE-128 (AD / meta-AD / PRF subset) and Merlin v1.0 transcripts,
// host side.  Replaces the `merlin` crate on the verifier's path (SURVEY.md sec
// 8(a) row a10; merlin.cool transcript protocol, STROBE v1.0.2 sec 5-6, FIPS 202).
// Every Fiat-Shamir challenge of r1cs::Verifier::verify is drawn through this.
#pragma once
#include "keccak.hpp"
#include "scalar.hpp"

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace zk {

class Strobe128 {
 public:
  explicit Strobe128(const char* protocol_label) {
    std::memset(st_, 0, sizeof st_);
    const uint8_t hdr[6] = {1, kRate + 2, 1, 0, 1, 96};
    for (int i = 0; i < 6; ++i) put(i, hdr[i]);
    const char* ver = "STROBEv1.0.2";
    for (int i = 0; i < 12; ++i) put(6 + i, (uint8_t)ver[i]);
    keccak_f1600(st_);
    meta_ad((const uint8_t*)protocol_label, std::strlen(protocol_label), false);
  }
  void meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(kM | kA, more); absorb(d, n); }
  void ad(const uint8_t* d, size_t n, bool more) { begin_op(kA, more); absorb(d, n); }
  // 50 state words + position + begin marker (seed of the device-side replay: transcript_tape.hpp, k_transcript)
  void export_state(uint32_t out[52]) const {
    for (int i = 0; i < 25; ++i) { out[2 * i] = (uint32_t)st_[i]; out[2 * i + 1] = (uint32_t)(st_[i] >> 32); }
    out[50] = pos_;
    out[51] = pos_begin_;
  }
  // KEY: overwrite the state with the key bytes (STROBE v1.0.2 sec 6; used by Merlin's TranscriptRng)
  void key(const uint8_t* d, size_t n, bool more) {
    begin_op(kA | kC, more);
    for (size_t i = 0; i < n; ++i) {
      put(pos_, d[i]);
      if (++pos_ == kRate) run_f();
    }
  }
  void prf(uint8_t* out, size_t n) {
    begin_op(kI | kA | kC, false);
    for (size_t i = 0; i < n; ++i) {
      out[i] = get(pos_);
      put(pos_, 0);
      if (++pos_ == kRate) run_f();
    }
  }

 private:
  static constexpr unsigned kRate = 166;
  static constexpr uint8_t kI = 1, kA = 2, kC = 4, kT = 8, kM = 16, kK = 32;
  uint64_t st_[25];
  unsigned pos_ = 0, pos_begin_ = 0;

  uint8_t get(unsigned i) const { return (uint8_t)(st_[i >> 3] >> (8 * (i & 7))); }
  void put(unsigned i, uint8_t b) { st_[i >> 3] = (st_[i >> 3] & ~(0xffULL << (8 * (i & 7)))) | ((uint64_t)b << (8 * (i & 7))); }
  void xor_in(unsigned i, uint8_t b) { st_[i >> 3] ^= (uint64_t)b << (8 * (i & 7)); }
  void run_f() {
    xor_in(pos_, (uint8_t)pos_begin_);
    xor_in(pos_ + 1, 0x04);
    xor_in(kRate + 1, 0x80);
    keccak_f1600(st_);
    pos_ = 0;
    pos_begin_ = 0;
  }
  void absorb(const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      xor_in(pos_, d[i]);
      if (++pos_ == kRate) run_f();
    }
  }
  void begin_op(uint8_t flags, bool more) {
    if (more) return;
    const uint8_t hdr[2] = {(uint8_t)pos_begin_, flags};
    pos_begin_ = pos_ + 1;
    absorb(hdr, 2);
    if ((flags & (kC | kK)) && pos_ != 0) run_f();
  }
};

class Transcript {
 public:
  explicit Transcript(const char* label) : s_("Merlin v1.0") { append_message("dom-sep", (const uint8_t*)label, std::strlen(label)); }
  void append_message(const char* label, const uint8_t* msg, size_t n) {
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    s_.meta_ad((const uint8_t*)label, std::strlen(label), false);
    s_.meta_ad(len, 4, true);
    s_.ad(msg, n, false);
  }
  void append_u64(const char* label, uint64_t x) {
    uint8_t b[8];
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(x >> (8 * i));
    append_message(label, b, 8);
  }
  void append_point(const char* label, const uint8_t p[32]) { append_message(label, p, 32); }
  void append_scalar(const char* label, const Scalar& s) {
    uint8_t b[32];
    s.to_bytes(b);
    append_message(label, b, 32);
  }
  void challenge_bytes(const char* label, uint8_t* out, size_t n) {
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    s_.meta_ad((const uint8_t*)label, std::strlen(label), false);
    s_.meta_ad(len, 4, true);
    s_.prf(out, n);
  }
  void export_state(uint32_t out[52]) const { s_.export_state(out); }
  // merlin::TranscriptRngBuilder / TranscriptRng (prover side: blinding factors).  Call on a COPY of
  // the transcript: rekey_with_witness_bytes for every secret, finalize with external randomness,
  // then fill.
  void rekey_with_witness(const char* label, const uint8_t* w, size_t n) {
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    s_.meta_ad((const uint8_t*)label, std::strlen(label), false);
    s_.meta_ad(len, 4, true);
    s_.key(w, n, false);
  }
  void finalize_rng(const uint8_t seed[32]) {
    s_.meta_ad((const uint8_t*)"rng", 3, false);
    s_.key(seed, 32, false);
  }
  void rng_fill(uint8_t* out, size_t n) {
    uint8_t len[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    s_.meta_ad(len, 4, false);
    s_.prf(out, n);
  }
  Scalar challenge_scalar(const char* label) {
    uint8_t b[64];
    challenge_bytes(label, b, 64);
    return Scalar::from_wide(b);
  }

 private:
  Strobe128 s_;
};

}  // namespace zk
