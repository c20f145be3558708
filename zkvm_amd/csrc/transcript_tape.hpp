// transcript_tape.hpp -- the Merlin transcript of r1cs::Verifier::verify as a tape.
//
// Every proof of one statement shape drives the SAME sequence of STROBE
// operations: the labels, the lengths, the framing bytes, every position inside
// the 166-byte rate and every place a Keccak-f permutation falls are functions
// of the shape alone.  So the host replays the framing ONCE per shape
// (TapeRecorder mirrors merlin.hpp's Strobe128 byte for byte) and records
//     XOR   word, value          constant bytes (labels, lengths, framing, padding), merged per word
//     DATA  source, offset, pos  n bytes of the proof / the commitments into the state at byte `pos`
//     PERM                       Keccak-f[1600]
//     CHAL  slot                 64 challenge bytes (state bytes 0..63, then zeroed) -> scalar slot
// and the device runs that tape for each transaction (prep_kernels.hpp,
// k_transcript): a four-way switch around ONE Keccak-f instead of a kernel with
// hundreds of inlined byte-wise absorb loops.
// (merlin.cool, STROBE v1.0.2 sec 5-7.)
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace zk {

enum : uint32_t { TAPE_XOR = 0, TAPE_DATA = 1, TAPE_PERM = 2, TAPE_CHAL = 3 };
enum : uint32_t { TAPE_SRC_COMMITMENTS = 0, TAPE_SRC_PROOF = 1 };

class TapeRecorder {
 public:
  static constexpr uint32_t R = 166;
  TapeRecorder(uint32_t pos, uint32_t pos_begin) : pos_(pos), pos_begin_(pos_begin) {}

  // Merlin framing (merlin.hpp Transcript::append_message / challenge_bytes)
  void append_data(const char* label, uint32_t src, uint32_t src_byte_off, uint32_t n) {
    meta(label, n);
    begin_op(kA);
    data(src, src_byte_off, n);
  }
  void append_const(const char* label, const void* msg, uint32_t n) {
    meta(label, n);
    begin_op(kA);
    const uint8_t* b = (const uint8_t*)msg;
    for (uint32_t i = 0; i < n; ++i) absorb_byte(b[i]);
  }
  void append_u64(const char* label, uint64_t x) {
    uint8_t b[8];
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)(x >> (8 * i));
    append_const(label, b, 8);
  }
  void challenge(const char* label, uint32_t slot) {
    meta(label, 64);
    begin_op(kI | kA | kC);           // PRF: always permutes (two framing bytes were just absorbed)
    emit(TAPE_CHAL, slot, 0, 0);      // state bytes 0..63 out and zeroed
    pos_ = 64;
  }
  // the words of the finished tape, four per operation
  std::vector<uint32_t> finish() {
    flush();
    return ops_;
  }

 private:
  static constexpr uint8_t kI = 1, kA = 2, kC = 4, kM = 16;
  void emit(uint32_t kind, uint32_t a, uint32_t b, uint32_t c) {
    ops_.push_back(kind); ops_.push_back(a); ops_.push_back(b); ops_.push_back(c);
  }
  void xor_byte(uint32_t i, uint32_t b) { pend_[i >> 2] ^= b << (8 * (i & 3)); }
  void flush() {
    for (auto& kv : pend_) if (kv.second) emit(TAPE_XOR, kv.first, kv.second, 0);
    pend_.clear();
  }
  void run_f() {
    xor_byte(pos_, pos_begin_);
    xor_byte(pos_ + 1, 0x04);
    xor_byte(R + 1, 0x80);
    flush();
    emit(TAPE_PERM, 0, 0, 0);
    pos_ = 0;
    pos_begin_ = 0;
  }
  void absorb_byte(uint32_t b) {
    xor_byte(pos_, b);
    if (++pos_ == R) run_f();
  }
  void begin_op(uint8_t flags) {
    const uint32_t old_begin = pos_begin_;
    pos_begin_ = pos_ + 1;
    absorb_byte(old_begin);
    absorb_byte(flags);
    if ((flags & (kC | 32)) && pos_ != 0) run_f();
  }
  void meta(const char* label, uint32_t n) {
    begin_op(kM | kA);
    for (const char* p = label; *p; ++p) absorb_byte((uint8_t)*p);
    for (int i = 0; i < 4; ++i) absorb_byte((n >> (8 * i)) & 0xff);   // continuation of the same meta-AD
  }
  void data(uint32_t src, uint32_t off, uint32_t n) {
    while (n) {
      const uint32_t take = n < R - pos_ ? n : R - pos_;
      emit(TAPE_DATA, src, off, pos_ | (take << 16));
      pos_ += take; off += take; n -= take;
      if (pos_ == R) run_f();
    }
  }

  uint32_t pos_, pos_begin_;
  std::map<uint32_t, uint32_t> pend_;
  std::vector<uint32_t> ops_;
};

// The transcript of bulletproofs' r1cs::Verifier::verify after the domain separator, for a
// statement with m commitments, the given second-phase challenge labels (Merlin labels, in the order the
// randomized constraints draw them) and an inner-product argument of k
// rounds over pn generators.  Proof bytes are addressed after the version byte: 32 i is field i of
// A_I1 A_O1 S1 A_I2 A_O2 S2 T_1 T_3 T_4 T_5 T_6 t_x t_x_blinding e_blinding L_0 R_0 ... a b.
// Challenge slots: y 0, z 1, u 2, x 3, w 4, second phase ch_fixed + j, u_j ch_fixed + n_chal2 + j.
// (Order and labels: r1cs_verifier.hpp, R1csVerifier::verify -- the same sequence on the host.)
inline std::vector<uint32_t> build_r1cs_verifier_tape(uint32_t pos, uint32_t pos_begin, uint32_t m,
                                                      const std::vector<std::string>& chal_label, uint32_t k, uint32_t pn,
                                                      uint32_t ch_fixed) {
  TapeRecorder rec(pos, pos_begin);
  const uint32_t n_chal2 = (uint32_t)chal_label.size();
  for (uint32_t i = 0; i < m; ++i) rec.append_data("V", TAPE_SRC_COMMITMENTS, 32 * i, 32);
  rec.append_u64("m", m);
  const char* first[3] = {"A_I1", "A_O1", "S1"};
  for (uint32_t i = 0; i < 3; ++i) rec.append_data(first[i], TAPE_SRC_PROOF, 32 * i, 32);
  if (n_chal2 == 0) {
    rec.append_const("dom-sep", "r1cs-1phase", 11);
  } else {
    rec.append_const("dom-sep", "r1cs-2phase", 11);
    for (uint32_t j = 0; j < n_chal2; ++j) rec.challenge(chal_label[j].c_str(), ch_fixed + j);
  }
  const char* second[3] = {"A_I2", "A_O2", "S2"};
  for (uint32_t i = 0; i < 3; ++i) rec.append_data(second[i], TAPE_SRC_PROOF, 32 * (3 + i), 32);
  rec.challenge("y", 0);
  rec.challenge("z", 1);
  const char* tl[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
  for (uint32_t i = 0; i < 5; ++i) rec.append_data(tl[i], TAPE_SRC_PROOF, 32 * (6 + i), 32);
  rec.challenge("u", 2);
  rec.challenge("x", 3);
  rec.append_data("t_x", TAPE_SRC_PROOF, 32 * 11, 32);
  rec.append_data("t_x_blinding", TAPE_SRC_PROOF, 32 * 12, 32);
  rec.append_data("e_blinding", TAPE_SRC_PROOF, 32 * 13, 32);
  rec.challenge("w", 4);
  rec.append_const("dom-sep", "ipp v1", 6);
  rec.append_u64("n", pn);
  for (uint32_t j = 0; j < k; ++j) {
    rec.append_data("L", TAPE_SRC_PROOF, 32 * (14 + 2 * j), 32);
    rec.append_data("R", TAPE_SRC_PROOF, 32 * (14 + 2 * j + 1), 32);
    rec.challenge("u", ch_fixed + n_chal2 + j);
  }
  return rec.finish();
}

// The same tape regrouped for the cooperative kernel (prep_kernels.hpp, k_transcript_coop: one Keccak state
// spread over a wavefront): a SEGMENT is everything up to and including one permutation --
//     [CHAL slot]  the 64 challenge bytes leave the state (always the first thing after a permutation)
//     absorb       constant bytes and proof / commitment bytes XORed into the state, in any order
//     [PERM]
// info[s]   bit 31: the segment ends with a permutation; bits 0..15: 1 + challenge slot, 0 = none
// consts    [n_seg][50] words XORed into the state
// map       [n_seg][200]: per state byte 0 = nothing, else 1 + index of the byte absorbed there, counted
//           in the concatenation (commitments, 32 m bytes | proof after its version byte)
struct CoopSegments {
  std::vector<uint32_t> info, consts;
  std::vector<uint16_t> map;
  uint32_t n_seg() const { return (uint32_t)info.size(); }
};

inline CoopSegments build_coop_segments(const std::vector<uint32_t>& tape, uint32_t m) {
  CoopSegments out;
  std::vector<uint32_t> c(50, 0);
  std::vector<uint16_t> mp(200, 0);
  uint32_t chal = 0;
  bool dirty = false;
  auto close = [&](bool perm) {
    out.info.push_back((perm ? 0x80000000u : 0u) | chal);
    out.consts.insert(out.consts.end(), c.begin(), c.end());
    out.map.insert(out.map.end(), mp.begin(), mp.end());
    c.assign(50, 0); mp.assign(200, 0); chal = 0; dirty = false;
  };
  for (size_t i = 0; i + 3 < tape.size(); i += 4) {
    const uint32_t kind = tape[i], a = tape[i + 1], b = tape[i + 2], w = tape[i + 3];
    if (kind == TAPE_XOR) {
      c[a] ^= b; dirty = true;
    } else if (kind == TAPE_DATA) {
      const uint32_t pos = w & 0xffffu, n = w >> 16;
      for (uint32_t q = 0; q < n; ++q) {
        const uint32_t idx = (a == TAPE_SRC_PROOF ? 32 * m : 0) + b + q;
        if (mp[pos + q] != 0 || idx >= 0xffffu) { out.info.clear(); return out; }   // not representable: caller falls back
        mp[pos + q] = (uint16_t)(idx + 1);
      }
      dirty = true;
    } else if (kind == TAPE_PERM) {
      close(true);
    } else {
      if (dirty || chal != 0 || a >= 0xffffu) { out.info.clear(); return out; }
      chal = a + 1;
    }
  }
  if (dirty || chal != 0) close(false);
  return out;
}

// Host interpreter of a tape (the reference semantics of k_transcript's loop; used by the CPU tests):
// state = 200 bytes, challenges[slot] = the 64 squeezed bytes.
template <typename Permute>
inline void run_tape_host(const std::vector<uint32_t>& tape, uint8_t state[200], const uint8_t* commitments,
                          const uint8_t* proof_after_version, Permute permute,
                          std::map<uint32_t, std::vector<uint8_t>>& challenges) {
  for (size_t i = 0; i + 3 < tape.size(); i += 4) {
    const uint32_t kind = tape[i], a = tape[i + 1], b = tape[i + 2], c = tape[i + 3];
    if (kind == TAPE_XOR) {
      for (int q = 0; q < 4; ++q) state[4 * a + q] ^= (uint8_t)(b >> (8 * q));
    } else if (kind == TAPE_DATA) {
      const uint8_t* src = a == TAPE_SRC_PROOF ? proof_after_version : commitments;
      const uint32_t pos = c & 0xffffu, n = c >> 16;
      for (uint32_t q = 0; q < n; ++q) state[pos + q] ^= src[b + q];
    } else if (kind == TAPE_PERM) {
      permute(state);
    } else {
      challenges[a].assign(state, state + 64);
      memset(state, 0, 64);
    }
  }
}

}  // namespace zk
