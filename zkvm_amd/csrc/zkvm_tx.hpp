// zkvm_tx.hpp -- the caller's side of the hot path (second half): a serialized ZkVM
// transaction -> what `Tx::verify` / `Verifier::verify_tx` hands to the proof system: the transaction ID, the keys the
// transaction signature must cover, and the statement + proof of its `cloak`.  Host code, as upstream's VM is.
//
// UNPINNED RECOLLECTION.  /root/reference holds no source and no specification: the wire format, the
// opcodes, the labels and the hashing below restate what the survey's appendix and the public ZkVM design notes
// describe, as far as they could be recalled; they are self-consistent across this file and the oracle
// (oracle/zkvm_tx.c, written separately) and NOT checkable against upstream here.  DESIGN.md sec 4.5 is the normative
// description of what this subset is; everything outside it is reported as "unsupported", never as "invalid".
//
//   Tx       := version:u64 | mintime_ms:u64 | maxtime_ms:u64 | n:u32 program[n] | R:32 s:32 | n:u32 proof[n]
//   program  := instruction*          (subset: a payment)
//       0x00 push:n:x   0x02 drop   0x03 dup:k   0x04 roll:k   0x06 var   0x18 cloak:m:n
//       0x1b input      0x1c output:k            0x20 signtx           (immediates: u32 little-endian)
//   Contract := anchor:32 | predicate:32 | k:u32 | item*      item := 0x00 n:u32 bytes[n]  |  0x02 qty:32 flavor:32
#pragma once
#include "merlin.hpp"
#include "scalar.hpp"

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace zk {
namespace zkvm {

enum TxStatus : uint8_t { TX_OK = 0, TX_INVALID = 1, TX_UNSUPPORTED = 2 };

// Bytes with inline room for the common case (a payment's four values, its few keys) and a heap fallback beyond: the
// statements of a chunk are then ONE allocation of the calling thread, instead of three small ones per transaction made
// on the worker threads and freed on the caller's (cross-thread frees: 2.5 ms of an 8192-transaction call).
template <size_t N>
class SmallBytes {
 public:
  SmallBytes() = default;
  SmallBytes(const SmallBytes& o) { assign(o.data(), o.size()); }
  SmallBytes& operator=(const SmallBytes& o) { if (this != &o) assign(o.data(), o.size()); return *this; }
  SmallBytes(SmallBytes&& o) noexcept : n_(o.n_), heap_(std::move(o.heap_)) { if (heap_.empty() && n_) std::memcpy(inl_, o.inl_, n_); o.n_ = 0; }
  SmallBytes& operator=(SmallBytes&& o) noexcept {
    if (this != &o) { n_ = o.n_; heap_ = std::move(o.heap_); if (heap_.empty() && n_) std::memcpy(inl_, o.inl_, n_); o.n_ = 0; }
    return *this;
  }
  void resize(size_t n) {                       // contents beyond the old size are zero
    if (n <= N && heap_.empty()) { if (n > n_) std::memset(inl_ + n_, 0, n - n_); n_ = n; return; }
    if (heap_.empty()) { heap_.assign(inl_, inl_ + n_); }
    heap_.resize(n, 0);
    n_ = n;
  }
  void assign(const uint8_t* p, size_t n) { heap_.clear(); n_ = 0; resize(n); if (n) std::memcpy(data(), p, n); }
  size_t size() const { return n_; }
  bool empty() const { return n_ == 0; }
  uint8_t* data() { return heap_.empty() ? inl_ : heap_.data(); }
  const uint8_t* data() const { return heap_.empty() ? inl_ : heap_.data(); }
  uint8_t& operator[](size_t i) { return data()[i]; }
  const uint8_t& operator[](size_t i) const { return data()[i]; }

 private:
  size_t n_ = 0;
  uint8_t inl_[N];
  std::vector<uint8_t> heap_;
};

struct TxStatement {
  TxStatus status = TX_INVALID;
  const char* why = "";
  uint64_t version = 0, mintime = 0, maxtime = 0;
  uint8_t txid[32] = {0};
  // the one cloak of the transaction: m inputs, n outputs, 64 bytes (qty, flavor commitments) per value, inputs first
  uint32_t n_in = 0, n_out = 0;
  SmallBytes<256> commitments;
  const uint8_t* proof = nullptr;
  size_t proof_len = 0;
  // the signature check  s B - R - sum_i (c a_i) X_i == identity  as terms of a multiscalar multiplication
  SmallBytes<160> sig_scalars, sig_points;        // 32 bytes each, same count
};

// A stack item REFERS to its bytes -- inside the transaction (pushed strings, and the contracts and payloads parsed out of
// them) or inside the statement's commitment buffer (the values a cloak leaves) -- and owns nothing: the VM of a payment
// moves a few dozen items around, and with owning items it spent a third of its time in the allocator.
struct Item {
  enum Kind : uint8_t { Data, Variable, Value, Contract } kind = Data;
  const uint8_t* p = nullptr;          // Data: the string; Variable: 32-byte commitment; Value: qty | flavor; Contract: its serialization
  size_t n = 0;
};

inline uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

// every hash below starts from a copy of its label's freshly initialised transcript (two Keccak-f saved per hash)
#define ZK_TX_TRANSCRIPT(var, label) static const Transcript var##_proto(label); Transcript var = var##_proto
inline void contract_id(const uint8_t* ser, size_t n, uint8_t id[32]) {
  ZK_TX_TRANSCRIPT(t, "ZkVM.contractid");
  t.append_message("contract", ser, n);
  t.challenge_bytes("id", id, 32);
}
inline void ratchet_anchor(const uint8_t old_anchor[32], uint8_t fresh[32]) {
  ZK_TX_TRANSCRIPT(t, "ZkVM.ratchet-anchor");
  t.append_message("old", old_anchor, 32);
  t.challenge_bytes("new", fresh, 32);
}

// Merkle root over the log entries (RFC 6962 split: the largest power of two below the count goes left), every hash a
// Merlin transcript: leaf = entry's own messages then "merkle.leaf"; node = "L", "R" then "merkle.node"
struct LogEntry {
  enum Kind : uint8_t { Header, Input, Output } kind;
  uint64_t a = 0, b = 0, c = 0;
  uint8_t id[32] = {0};
};
inline void merkle_root(const Transcript& fresh, const LogEntry* e, size_t n, uint8_t out[32]) {
  Transcript t = fresh;
  if (n == 0) { t.challenge_bytes("merkle.empty", out, 32); return; }
  if (n == 1) {
    switch (e->kind) {
      case LogEntry::Header: t.append_u64("tx.version", e->a); t.append_u64("tx.mintime", e->b); t.append_u64("tx.maxtime", e->c); break;
      case LogEntry::Input: t.append_message("input", e->id, 32); break;
      case LogEntry::Output: t.append_message("output", e->id, 32); break;
    }
    t.challenge_bytes("merkle.leaf", out, 32);
    return;
  }
  size_t k = 1;
  while (2 * k < n) k *= 2;
  uint8_t l[32], r[32];
  merkle_root(fresh, e, k, l);
  merkle_root(fresh, e + k, n - k, r);
  t.append_message("L", l, 32);
  t.append_message("R", r, 32);
  t.challenge_bytes("merkle.node", out, 32);
}

// parses a serialized contract; false: malformed.  items (Data or Value) refer into `p`
inline bool parse_contract(const uint8_t* p, size_t n, const uint8_t*& predicate, std::vector<Item>* items, bool& unsupported) {
  if (n < 68) return false;
  predicate = p + 32;
  const uint32_t k = rd32(p + 64);
  size_t pos = 68;
  for (uint32_t i = 0; i < k; ++i) {
    if (pos >= n) return false;
    const uint8_t type = p[pos++];
    Item it;
    if (type == 0x00) {
      if (n - pos < 4) return false;
      const uint32_t len = rd32(p + pos);
      pos += 4;
      if (n - pos < len) return false;
      it.kind = Item::Data;
      it.p = p + pos; it.n = len;
      pos += len;
    } else if (type == 0x02) {
      if (n - pos < 64) return false;
      it.kind = Item::Value;
      it.p = p + pos; it.n = 64;
      pos += 64;
    } else if (type == 0x01) {
      unsupported = true;      // a program item: outside the subset
      return false;
    } else {
      return false;
    }
    if (items) items->push_back(it);
  }
  return pos == n;
}

inline void serialize_contract(const uint8_t anchor[32], const uint8_t predicate[32], const Item* items, size_t n_items, std::vector<uint8_t>& out) {
  out.assign(anchor, anchor + 32);
  out.insert(out.end(), predicate, predicate + 32);
  const uint32_t k = (uint32_t)n_items;
  for (int b = 0; b < 4; ++b) out.push_back((uint8_t)(k >> (8 * b)));
  for (size_t q = 0; q < n_items; ++q) {
    const Item& it = items[q];
    if (it.kind == Item::Value) {
      out.push_back(0x02);
      out.insert(out.end(), it.p, it.p + it.n);
    } else {
      out.push_back(0x00);
      const uint32_t len = (uint32_t)it.n;
      for (int b = 0; b < 4; ++b) out.push_back((uint8_t)(len >> (8 * b)));
      out.insert(out.end(), it.p, it.p + it.n);
    }
  }
}

// Runs the transaction: parse, VM, transaction ID, signature equation.  The statement refers into `tx` (proof bytes).
inline TxStatement tx_prepare(const uint8_t* tx, size_t len) {
  TxStatement st;
  auto fail = [&st](TxStatus s, const char* why) { st.status = s; st.why = why; return st; };
  if (len < 24 + 4) return fail(TX_INVALID, "truncated header");
  st.version = rd64(tx); st.mintime = rd64(tx + 8); st.maxtime = rd64(tx + 16);
  size_t pos = 24;
  const uint32_t prog_len = rd32(tx + pos);
  pos += 4;
  if (len - pos < prog_len) return fail(TX_INVALID, "truncated program");
  const uint8_t* prog = tx + pos;
  pos += prog_len;
  if (len - pos < 64 + 4) return fail(TX_INVALID, "truncated signature");
  const uint8_t* sig = tx + pos;
  pos += 64;
  const uint32_t proof_len = rd32(tx + pos);
  pos += 4;
  if (len - pos != proof_len) return fail(TX_INVALID, "proof length does not match the transaction's");
  st.proof = tx + pos;
  st.proof_len = proof_len;
  if (st.version != 1) return fail(TX_UNSUPPORTED, "transaction version");
  if (st.mintime > st.maxtime) return fail(TX_INVALID, "mintime after maxtime");

  // scratch of the calling thread, reused from transaction to transaction
  static thread_local std::vector<Item> stack, payload;
  static thread_local std::vector<LogEntry> log;
  static thread_local std::vector<const uint8_t*> keys;
  static thread_local std::vector<uint8_t> ser;
  stack.clear(); log.clear(); keys.clear();
  LogEntry hdr; hdr.kind = LogEntry::Header; hdr.a = st.version; hdr.b = st.mintime; hdr.c = st.maxtime;
  log.push_back(hdr);
  bool have_anchor = false, cloaked = false;
  uint8_t last_anchor[32] = {0};
  size_t pc = 0;
  auto imm32 = [&](uint32_t& v) { if (prog_len - pc < 4) return false; v = rd32(prog + pc); pc += 4; return true; };
  while (pc < prog_len) {
    const uint8_t op = prog[pc++];
    switch (op) {
      case 0x00: {   // push:n:x
        uint32_t n;
        if (!imm32(n) || prog_len - pc < n) return fail(TX_INVALID, "push runs past the program");
        Item it; it.kind = Item::Data; it.p = prog + pc; it.n = n;
        pc += n;
        stack.push_back(it);
        break;
      }
      case 0x02:     // drop
        if (stack.empty()) return fail(TX_INVALID, "stack underflow");
        if (stack.back().kind == Item::Value || stack.back().kind == Item::Contract) return fail(TX_INVALID, "drop of a value or a contract");
        stack.pop_back();
        break;
      case 0x03: {   // dup:k
        uint32_t k;
        if (!imm32(k) || k >= stack.size()) return fail(TX_INVALID, "dup out of range");
        const Item src = stack[stack.size() - 1 - k];
        if (src.kind == Item::Value || src.kind == Item::Contract) return fail(TX_INVALID, "dup of a value or a contract");
        stack.push_back(src);
        break;
      }
      case 0x04: {   // roll:k
        uint32_t k;
        if (!imm32(k) || k >= stack.size()) return fail(TX_INVALID, "roll out of range");
        const Item it = stack[stack.size() - 1 - k];
        stack.erase(stack.end() - 1 - k);
        stack.push_back(it);
        break;
      }
      case 0x06: {   // var
        if (stack.empty() || stack.back().kind != Item::Data || stack.back().n != 32) return fail(TX_INVALID, "var needs a 32-byte commitment");
        stack.back().kind = Item::Variable;
        break;
      }
      case 0x18: {   // cloak:m:n
        uint32_t m, n;
        if (!imm32(m) || !imm32(n)) return fail(TX_INVALID, "cloak immediates");
        if (cloaked) return fail(TX_UNSUPPORTED, "more than one cloak per transaction");
        if (m == 0 || n == 0 || m > 64 || n > 64) return fail(TX_UNSUPPORTED, "cloak arity");
        if (stack.size() < (size_t)m + 2 * (size_t)n) return fail(TX_INVALID, "stack underflow");
        st.commitments.resize(64 * ((size_t)m + n));      // inputs, then outputs (the values pushed below refer into it)
        uint8_t* ins = st.commitments.data();
        uint8_t* outs = ins + 64 * (size_t)m;
        for (uint32_t j = n; j-- > 0;) {     // .. q_j f_j on top
          for (int half = 1; half >= 0; --half) {
            if (stack.back().kind != Item::Variable) return fail(TX_INVALID, "cloak outputs must be variables");
            std::memcpy(&outs[64 * j + 32 * half], stack.back().p, 32);
            stack.pop_back();
          }
        }
        for (uint32_t i = m; i-- > 0;) {
          if (stack.back().kind != Item::Value) return fail(TX_INVALID, "cloak inputs must be values");
          std::memcpy(&ins[64 * i], stack.back().p, 64);
          stack.pop_back();
        }
        st.n_in = m; st.n_out = n;
        for (uint32_t j = 0; j < n; ++j) { Item v; v.kind = Item::Value; v.p = &outs[64 * j]; v.n = 64; stack.push_back(v); }
        cloaked = true;
        break;
      }
      case 0x1b: {   // input
        if (stack.empty() || stack.back().kind != Item::Data) return fail(TX_INVALID, "input needs a serialized contract");
        const uint8_t* pred;
        bool unsupported = false;
        const Item c_ser = stack.back();
        stack.pop_back();
        if (!parse_contract(c_ser.p, c_ser.n, pred, nullptr, unsupported))
          return fail(unsupported ? TX_UNSUPPORTED : TX_INVALID, "malformed contract");
        LogEntry e; e.kind = LogEntry::Input;
        contract_id(c_ser.p, c_ser.n, e.id);
        log.push_back(e);
        std::memcpy(last_anchor, e.id, 32);
        have_anchor = true;
        Item c = c_ser; c.kind = Item::Contract;
        stack.push_back(c);
        break;
      }
      case 0x1c: {   // output:k
        uint32_t k;
        if (!imm32(k)) return fail(TX_INVALID, "output immediate");
        if (stack.size() < (size_t)k + 1) return fail(TX_INVALID, "stack underflow");
        if (stack.back().kind != Item::Data || stack.back().n != 32) return fail(TX_INVALID, "output needs a 32-byte predicate");
        if (!have_anchor) return fail(TX_INVALID, "output before any input: no anchor");
        const uint8_t* pred = stack.back().p;
        stack.pop_back();
        const Item* items = stack.data() + (stack.size() - k);
        for (uint32_t q = 0; q < k; ++q) if (items[q].kind != Item::Data && items[q].kind != Item::Value) return fail(TX_INVALID, "output payload must be data or values");
        uint8_t anchor[32];
        ratchet_anchor(last_anchor, anchor);
        std::memcpy(last_anchor, anchor, 32);
        serialize_contract(anchor, pred, items, k, ser);
        stack.resize(stack.size() - k);
        LogEntry e; e.kind = LogEntry::Output;
        contract_id(ser.data(), ser.size(), e.id);
        log.push_back(e);
        break;
      }
      case 0x20: {   // signtx
        if (stack.empty() || stack.back().kind != Item::Contract) return fail(TX_INVALID, "signtx needs a contract");
        const uint8_t* pred;
        bool unsupported = false;
        const Item c_ser = stack.back();
        stack.pop_back();
        payload.clear();
        if (!parse_contract(c_ser.p, c_ser.n, pred, &payload, unsupported)) return fail(TX_INVALID, "malformed contract");
        keys.push_back(pred);
        for (const Item& it : payload) stack.push_back(it);
        break;
      }
      default:
        return fail(TX_UNSUPPORTED, "instruction outside the payment subset");
    }
  }
  if (!stack.empty()) return fail(TX_INVALID, "stack not empty at the end");
  if (!cloaked) return fail(TX_UNSUPPORTED, "no cloak: nothing for the proof system");
  if (keys.empty()) return fail(TX_INVALID, "no key signs the transaction");
  static const Transcript txid_proto("ZkVM.txid");
  merkle_root(txid_proto, log.data(), log.size(), st.txid);

  // signature: X = sum a_i X_i (MuSig key aggregation), c = H(txid, X, R);  s B == R + c X
  Scalar s;
  if (!Scalar::from_canonical(sig + 32, s)) return fail(TX_INVALID, "signature scalar not canonical");
  ZK_TX_TRANSCRIPT(agg, "Musig.aggregated-key");
  agg.append_u64("n", keys.size());
  for (const uint8_t* k : keys) agg.append_point("X", k);
  static thread_local std::vector<Scalar> a;
  a.resize(keys.size());
  for (size_t i = 0; i < keys.size(); ++i) {
    Transcript ti = agg;
    ti.append_u64("i", i);
    a[i] = ti.challenge_scalar("a_i");
  }
  // the aggregated key as an encoding is not needed by the equation, but it is what the challenge binds: it has to be
  // computed, which needs the group -- the caller (device or test library) supplies it through `aggregate`
  st.sig_scalars.resize(32 * (2 + keys.size()));
  st.sig_points.resize(32 * (2 + keys.size()));
  st.status = TX_OK;
  // slots: [0] B (filled by the caller: scalar s), [1] R (scalar -1), [2 + i] X_i (scalar -c a_i, c filled in by finish_signature)
  s.to_bytes(&st.sig_scalars[0]);
  (-Scalar::one()).to_bytes(&st.sig_scalars[32]);
  std::memcpy(&st.sig_points[32], sig, 32);
  for (size_t i = 0; i < keys.size(); ++i) {
    a[i].to_bytes(&st.sig_scalars[32 * (2 + i)]);          // a_i for now
    std::memcpy(&st.sig_points[32 * (2 + i)], keys[i], 32);
  }
  return st;
}

// second half of the signature preparation, once the aggregated key X = sum a_i X_i is known as an encoding
// (one small multiscalar multiplication per transaction: zkgpu_msm_batch on the device, host_rows in the CPU tests)
inline void tx_finish_signature(TxStatement& st, const uint8_t basepoint[32], const uint8_t agg_key[32]) {
  ZK_TX_TRANSCRIPT(t, "ZkVM.signtx");
  t.append_message("txid", st.txid, 32);
  t.append_message("dom-sep", (const uint8_t*)"schnorr-signature v1", 20);
  t.append_point("X", agg_key);
  t.append_point("R", &st.sig_points[32]);
  const Scalar c = t.challenge_scalar("c");
  std::memcpy(&st.sig_points[0], basepoint, 32);
  const size_t n_keys = st.sig_scalars.size() / 32 - 2;
  for (size_t i = 0; i < n_keys; ++i) {
    Scalar ai;
    Scalar::from_canonical(&st.sig_scalars[32 * (2 + i)], ai);
    (-(c * ai)).to_bytes(&st.sig_scalars[32 * (2 + i)]);
  }
}

}  // namespace zkvm
}  // namespace zk
