// tx_call.hpp -- the SCHEDULING of one zkgpu_tx_verify_batch call (upstream: zkvm
// `Verifier::verify_tx` over a block, as recalled -- no file:line exists under /root/reference): chunks, stages, the two
// host threads and everything they share, with the device behind an interface.
//
// Until round 3 this was one 480-line function of fifteen capturing lambdas inside session.hpp (the
// round-2 and round-3 advisors each found a hazard in it).  It is a class now, and it does not know HIP: the same code is
// driven (a) by the verifier in libzkgpu (session.hpp: GpuTxDevice) and (b) in libzkhost by a CPU stand-in whose
// "device" finishes its work on threads of its own after random delays (hostlib.cpp: zkhost_txcall_selftest), which the
// sanitizer tier runs under ThreadSanitizer -- the flags, the ring, the stage cuts and the hand-overs between the two
// threads are exercised on the CPU with no GPU in sight.
//
// A call is cut into equal chunks of at most 8192 transactions (4096 when the call is longer than 16 384; an override for
// tests).  Two threads:
//   the STAGING thread (made for the call; its loops run on the worker pool): per chunk a first VM pass as far as the
//     signature's keys and their MuSig coefficients (or, for a call that is ONE chunk, the stack machine alone first: the
//     proofs go out before anything is hashed), the rows (a_i, X_i) of the aggregated keys, the chunk's cloak statements
//     gathered for the device (TxDevice::proofs_stage); then, per chunk, the second pass -- everything else the VM hashes --
//     and, whenever a run of chunks has its transaction IDs made AND its aggregated keys back, the signature transcripts and
//     the rows of the equations s B - R - sum (c a_i) X_i;
//   the CALLING thread: everything that talks to the device, never waiting for one thing while another could be queued:
//     key stages (two slots, alternating), proofs (a ring of RING staging areas), signature stages (two slots).
// accept = the VM accepted & the keys decode & the signature holds & the proof verifies; fail-closed in both outputs.
#pragma once
#include "host_pool.hpp"
#include "zkvm_tx.hpp"

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace zk {
namespace zkvm {

struct TxProofSource { uint32_t n_in, n_out; const uint8_t* com; const uint8_t* proof; uint64_t proof_len; };

// What a call needs of the device.  Status codes are the library's (0 = ok).  `slot` is 0 or 1: two stages of a kind may be
// in flight, and TxCall never reuses a slot before it has collected what the slot held.
class TxDevice {
 public:
  virtual ~TxDevice() {}
  virtual const uint8_t* basepoint() = 0;                                   // 32 bytes: the encoding of B
  // aggregated keys: rows of (a_i, X_i) -> per row the encoding of sum a_i X_i (values) and "every key decodes" (ok bits)
  virtual int keys_enqueue(int slot, const uint8_t* scalars, const uint8_t* points, const uint64_t* offsets, size_t rows) = 0;
  virtual bool keys_done(int slot) = 0;                                     // never blocks
  virtual int keys_collect(int slot, uint8_t* ok_bits, uint8_t* values) = 0;
  // cloak proofs of one chunk.  proofs_stage runs on the STAGING thread (host work: grouping by shape, gathering into the
  // ring slot's staging area); start / finish / release on the calling thread.  A handle that was staged is released
  // exactly once -- by proofs_finish, or by proofs_release if it was never started.
  virtual int proofs_stage(size_t ring_slot, size_t n, const TxProofSource* src, int host_threads, void** handle, std::string* err) = 0;
  virtual int proofs_start(size_t ring_slot, void* handle) = 0;
  virtual bool proofs_done(void* handle) = 0;                               // never blocks: would proofs_finish return at once?
  virtual int proofs_finish(void* handle, uint8_t* accept_bits) = 0;
  virtual void proofs_release(void* handle) = 0;
  // signature equations: per row dynamic terms (scalars, points) + one term on the basepoint's table -> "is the identity"
  virtual int sigs_enqueue(int slot, size_t rows, const uint8_t* dyn_scalars, const uint8_t* dyn_points, const uint64_t* dyn_offsets,
                           const uint8_t* base_scalars) = 0;
  virtual bool sigs_done(int slot) = 0;                                     // never blocks
  virtual int sigs_collect(int slot, uint8_t* bits) = 0;
  virtual std::string last_error() = 0;
};

class TxCall {
 public:
  static constexpr size_t RING = 6;                      // staging areas for the proofs of chunks in flight
  enum : int { OK = 0, ENOMEM_ = -4 };                   // (ZKGPU_OK / ZKGPU_ENOMEM: this header does not see zkgpu.h)

  // store: what the VM leaves per transaction, kept by the caller between calls (fresh memory costs a page fault per 4 KB);
  // at most `kept` entries of it are used, the rest of a longer call lives in the call.
  TxCall(TxDevice& dev, std::vector<TxStatement>& store, size_t kept, size_t batch, const uint8_t* txs, const uint64_t* tx_offsets,
         int host_threads, size_t chunk_override, uint8_t* accept_bitmap, uint8_t* status, int n_slots = 2)
      : dev_(dev), store_(store), batch_(batch), threads_(host_threads), accept_(accept_bitmap),
        status_(status), timing_(getenv("ZKGPU_PROVER_TIMING") != nullptr), t00_(now()), n_slots_((size_t)std::max(1, std::min(2, n_slots))) {
    ptr_.resize(batch); len_.resize(batch);
    for (size_t i = 0; i < batch; ++i) { ptr_[i] = txs + tx_offsets[i]; len_[i] = (size_t)(tx_offsets[i + 1] - tx_offsets[i]); }
    plan(chunk_override, kept);
  }
  // several callers' transactions as ONE call (zkgpu_tx_verify_submit: calls in flight are merged as tickets are): the
  // pieces in order, transaction i of the merged call = the i-th transaction counted through them
  struct Piece { const uint8_t* txs; const uint64_t* tx_offsets; size_t batch; };
  TxCall(TxDevice& dev, std::vector<TxStatement>& store, size_t kept, const std::vector<Piece>& pieces, int host_threads,
         size_t chunk_override, uint8_t* accept_bitmap, uint8_t* status, int n_slots = 2)
      : dev_(dev), store_(store), batch_(total_of(pieces)), threads_(host_threads), accept_(accept_bitmap), status_(status),
        timing_(getenv("ZKGPU_PROVER_TIMING") != nullptr), t00_(now()), n_slots_((size_t)std::max(1, std::min(2, n_slots))) {
    ptr_.reserve(batch_); len_.reserve(batch_);
    for (const Piece& pc : pieces)
      for (size_t i = 0; i < pc.batch; ++i) { ptr_.push_back(pc.txs + pc.tx_offsets[i]); len_.push_back((size_t)(pc.tx_offsets[i + 1] - pc.tx_offsets[i])); }
    plan(chunk_override, kept);
  }
  ~TxCall() { stop_stager(); }
  TxCall(const TxCall&) = delete;
  TxCall& operator=(const TxCall&) = delete;

  // -> 0, or the first error (both outputs then still read "nothing accepted"); error_text(): what it was
  int run() {
    const int rc = start();
    if (rc != OK) return rc;
    while (!done()) {
      if (step()) continue;
      std::unique_lock<std::mutex> lk(hm_);
      const double t0 = now();
#if defined(__SANITIZE_THREAD__)
      // (gcc 11's ThreadSanitizer does not intercept pthread_cond_clockwait, which wait_for on the steady clock becomes with
      // glibc >= 2.30: it then misses the unlock inside the wait and reports a "double lock".  Same nap on the system clock.)
      hcv_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::microseconds(50));
#else
      hcv_.wait_for(lk, std::chrono::microseconds(50));
#endif
      t_wait_host_ += now() - t0;
    }
    return finish();
  }
  // The same in pieces, for a caller that drives SEVERAL calls from one thread (the engine of zkgpu_tx_verify_submit): start()
  // makes the staging thread; step() does whatever can be done now without blocking and says whether anything was; done():
  // nothing is left to queue (or the call has failed); finish() collects what is in flight and writes the verdicts.
  // on_news: called by the staging thread whenever it has published something (to wake a caller that naps elsewhere).
  void set_on_news(std::function<void()> f) { on_news_ = std::move(f); }
  int start() {
    try {
      stager_ = std::thread([this] {
        try {
          staging_main();
        } catch (...) {                                  // std::bad_alloc in practice: an error for the call, not the end of the process
          std::lock_guard<std::mutex> lk(hm_);
          stager_failed_ = true;
          hcv_.notify_all();
        }
        if (on_news_) on_news_();
      });
    } catch (...) {                                      // no thread to be had: an error for the call as well
      error_ = "the staging thread of the call could not be started";
      rc_ = ENOMEM_;
      finished_ = true;
      return ENOMEM_;
    }
    return OK;
  }
  bool done() const { return finished_; }
  // after done(): has the device finished everything this call still has in flight, i.e. would finish() return without
  // waiting?  (A caller that drives several calls asks before it calls finish(), so that it never sleeps inside one call
  // while another has work to queue.)  Never blocks.
  bool settled() {
    for (size_t s = 0; s < seg_.size(); ++s) if (seg_[s].pending && !dev_.keys_done((int)(s % n_slots_))) return false;
    const size_t n_sig = n_sig_made_locked();
    for (size_t s = 0; s < n_sig; ++s) if (sig_stages_[s]->pending && !dev_.sigs_done((int)(s % n_slots_))) return false;
    for (const auto& k : chunks_) if (k->handle && k->started && !dev_.proofs_done(k->handle)) return false;
    return true;
  }
  int finish() {
    stop_stager();
    if (timing_ && rc_ == OK) {                          // (timing only: poll, so that the marks say which chain ends the call)
      std::vector<uint8_t> seen_s(sig_stages_.size(), 0);
      const double t_end = now() + 1.0;
      while (!settled() && now() < t_end) {
        device_marks();
        for (size_t q = 0; q < n_sig_made_locked(); ++q)
          if (!seen_s[q] && sig_stages_[q]->pending && dev_.sigs_done((int)(q % n_slots_))) { seen_s[q] = 1; mark("signatures done on the device, stage", q); }
      }
      device_marks();
    }
    // (after an error: nothing is left pending on the device)
    for (size_t s = 0; s < seg_.size(); ++s) keys_collect(s);
    for (size_t s = 0; s < n_sig_made_locked(); ++s) sigs_collect(s);
    for (size_t c = 0; c < chunks_.size(); ++c) proofs_collect(*chunks_[c]);
    report();
    if (rc_ != OK) return rc_;
    verdicts();
    return OK;
  }
  const std::string& error_text() const { return error_; }
  size_t n_chunks() const { return chunks_.size(); }
  size_t n_sig_stages_planned() const { return sig_plan_.size(); }

 private:
  struct Chunk {
    size_t lo = 0, n = 0, index = 0;                     // transactions [lo, lo + n) of the call; which chunk
    size_t g0 = 0;                                       // live transactions of the call before this chunk
    std::vector<size_t> live;                            // positions in the chunk the VM accepted
    void* handle = nullptr;                              // its staged proofs
    bool started = false;
    std::vector<uint8_t> pbits;
    int stage_rc = 0;
    std::string stage_err;
  };
  struct Segment {                                       // the key stage of one chunk
    size_t g_lo = 0, g_hi = 0;                           // live transactions of the call (positions in live_all_)
    std::vector<uint64_t> koff;                          // rows (a_i, X_i) per live transaction
    std::vector<uint8_t> ksc, kpt, kok;
    bool pending = false;
  };
  struct SigStage {                                      // the signature equations of a run of chunks
    size_t first = 0, last = 0;                          // chunks [first, last)
    std::vector<size_t> keyed;                           // global live indices of the transactions whose keys all decode
    std::vector<uint64_t> soff;
    std::vector<uint8_t> ssc, spt, sst, bits;
    bool pending = false;
  };
  struct Span { size_t first, last; };

  static size_t total_of(const std::vector<Piece>& pieces) { size_t n = 0; for (const Piece& p : pieces) n += p.batch; return n; }
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void mark(const char* what, size_t i) const { if (timing_) fprintf(stderr, "    %7.3f ms  %s %zu\n", (now() - t00_) * 1e3, what, i); }
  TxStatement& statement(size_t i) { return i < store_.size() ? store_[i] : beyond_[i - store_.size()]; }
  void note(int rc, const std::string& what) { if (rc != OK && rc_ == OK) { rc_ = rc; error_ = what; } }

  // ---- the plan: everything that depends on the SIZE of the call alone, so that the same call made again has stages of the
  // same sizes and the device side's buffers, grown once, are never grown again
  void plan(size_t chunk_override, size_t kept) {
    std::vector<size_t> cuts{0};
    const size_t chunk = chunk_override ? chunk_override : (batch_ <= 16384 ? 8192 : 4096);
    const size_t parts = (batch_ + chunk - 1) / chunk;
    for (size_t q = 1; q < parts; ++q) cuts.push_back(chunk_override ? std::min(batch_, q * chunk) : batch_ * q / parts);
    cuts.push_back(batch_);
    // (multiples of eight: the groups of eight the VM hashes in lockstep are then the same in both passes, and none
    // straddles the end of the statement store)
    for (size_t q = 1; q + 1 < cuts.size(); ++q) cuts[q] = std::max(cuts[q - 1], cuts[q] & ~(size_t)7);
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    if (cuts.size() < 2) cuts = {0, batch_};
    const size_t n = cuts.size() - 1;
    chunks_.resize(n);
    for (size_t c = 0; c < n; ++c) {
      chunks_[c].reset(new Chunk());
      chunks_[c]->lo = cuts[c]; chunks_[c]->n = cuts[c + 1] - cuts[c]; chunks_[c]->index = c;
    }
    seg_.resize(n);
    // signature stages: runs of chunks of ~10 000 transactions in all, the last run's last chunk on its own
    for (size_t c = 0; c < n;) {
      const size_t first = c;
      size_t count = 0;
      while (c < n && (count == 0 || count + chunks_[c]->n <= 10240 + 1536)) { count += chunks_[c]->n; ++c; }
      runs_.push_back({first, c});
      if (c == n && c - first > 1) { sig_plan_.push_back({first, c - 1}); sig_plan_.push_back({c - 1, c}); }
      else sig_plan_.push_back({first, c});
    }
    sig_stages_.reserve(sig_plan_.size());               // the calling thread indexes it while the staging thread appends: it never moves
    staged_.assign(n, 0); arena_free_.assign(n, 0); key_rows_.assign(n, 0); keys_arrived_.assign(n, 0);
    const size_t k8 = kept & ~(size_t)7;                 // (a multiple of eight: see the chunk boundaries)
    if (store_.size() > k8) { store_.resize(k8); store_.shrink_to_fit(); }
    if (store_.size() < std::min(batch_, k8)) store_.resize(std::min(batch_, k8));
    beyond_.resize(batch_ > store_.size() ? batch_ - store_.size() : 0);
    live_all_.reserve(batch_);
    agg_.assign(32 * std::max<size_t>(batch_, 1), 0);
    key_ok_.assign(batch_, 0);
    // A call that is ONE chunk sends its proofs first: theirs is the longest chain, and what they need is known once the VM's
    // stack machine has run, before anything is hashed.  A longer call keeps "keys, then proofs" per chunk: with proofs of
    // earlier chunks on the device, key kernels queued behind a chunk's proofs wait too long.
    proofs_first_ = n == 1;
  }

  // ================================================ staging thread ================================================
  void vm_pass(Chunk& k, uint8_t only) {                // the VM over the chunk, hashing the jobs of `only` alone
    const double t0 = now();
    const size_t t_lo = k.lo, t_hi = k.lo + k.n;
    host_parallel((t_hi - t_lo + 7) / 8, threads_, [&](size_t g) {
      const uint8_t* p[8];
      size_t l[8];
      const size_t first = t_lo + 8 * g, cnt = std::min<size_t>(8, t_hi - first);
      for (size_t q = 0; q < cnt; ++q) { p[q] = ptr_[first + q]; l[q] = len_[first + q]; }
      // (consecutive statements: the store's, or the overflow's -- a group of eight never straddles the two)
      tx_prepare_many(p, l, &statement(first), cnt, true, only);
    });
    std::lock_guard<std::mutex> lk(hm_);
    (only == ALL_PROTOS ? t_vm_ : t_keys_host_) += now() - t0;
  }
  void scan(Chunk& k, Segment& sg) {                    // which transactions the VM accepts
    sg.g_lo = k.g0 = live_all_.size();
    for (size_t i = 0; i < k.n; ++i) {
      const TxStatement& t = statement(k.lo + i);
      if (status_ && t.status == TX_UNSUPPORTED) status_[k.lo + i] = TX_UNSUPPORTED;     // (TX_OK is written at the very end only)
      if (t.status == TX_OK) { k.live.push_back(i); live_all_.push_back(k.lo + i); }
    }
    k.pbits.assign((k.live.size() + 7) / 8 + 1, 0);
    sg.g_hi = live_all_.size();
  }
  void key_rows_out(size_t ci, Segment& sg) {           // rows (a_i, X_i) of the chunk's aggregated keys -> the calling thread
    const double t0 = now();
    const size_t nl = sg.g_hi - sg.g_lo;
    sg.kok.assign((nl + 7) / 8 + 1, 0);
    sg.koff.assign(nl + 1, 0);
    for (size_t j = 0; j < nl; ++j) sg.koff[j + 1] = sg.koff[j] + statement(live_all_[sg.g_lo + j]).sig_scalars.size() / 32 - 2;
    sg.ksc.resize(32 * sg.koff.back()); sg.kpt.resize(32 * sg.koff.back());
    host_parallel(nl, threads_, [&](size_t j) {
      const TxStatement& t = statement(live_all_[sg.g_lo + j]);
      memcpy(sg.ksc.data() + 32 * sg.koff[j], t.sig_scalars.data() + 64, t.sig_scalars.size() - 64);
      memcpy(sg.kpt.data() + 32 * sg.koff[j], t.sig_points.data() + 64, t.sig_points.size() - 64);
    });
    std::lock_guard<std::mutex> lk(hm_);
    t_keys_host_ += now() - t0;
    key_rows_[ci] = 1;
    news();
  }
  bool gather(size_t ci, Chunk& k) {                    // the chunk's cloak statements into its staging area; false: the call is over
    {
      std::unique_lock<std::mutex> lk(hm_);
      if (ci >= RING) hcv_.wait(lk, [&] { return quit_ || arena_free_[ci - RING]; });
      if (quit_) return false;
    }
    const double t1 = now();
    const size_t nl = k.live.size();
    if (nl) {
      std::vector<TxProofSource> src(nl);
      for (size_t q = 0; q < nl; ++q) {
        const TxStatement& t = statement(k.lo + k.live[q]);
        src[q] = TxProofSource{t.n_in, t.n_out, t.commitments.data(), t.proof, t.proof_len};
      }
      k.stage_rc = dev_.proofs_stage(ci % RING, nl, src.data(), threads_, &k.handle, &k.stage_err);
    }
    const double t2 = now();
    if (timing_) fprintf(stderr, "    staging thread, chunk %zu: gather %.3f ms\n", ci, (t2 - t1) * 1e3);
    std::lock_guard<std::mutex> lk(hm_);
    t_stage_host_ += t2 - t1;
    staged_[ci] = 1;
    news();
    return true;
  }
  void sig_rows(SigStage& sg) {                         // transcripts + rows of the equations of chunks [first, last)
    const size_t g_lo = chunks_[sg.first]->g0, g_hi = chunks_[sg.last - 1]->g0 + chunks_[sg.last - 1]->live.size();
    sg.keyed.clear();
    for (size_t g = g_lo; g < g_hi; ++g) if (key_ok_[g]) sg.keyed.push_back(g);
    const size_t ns = sg.keyed.size();
    sg.bits.assign((ns + 7) / 8 + 1, 0);
    if (ns == 0) return;
    sg.soff.assign(ns + 1, 0);
    for (size_t q = 0; q < ns; ++q) sg.soff[q + 1] = sg.soff[q] + statement(live_all_[sg.keyed[q]]).sig_scalars.size() / 32 - 1;
    sg.ssc.resize(32 * sg.soff.back()); sg.spt.resize(32 * sg.soff.back());
    sg.sst.resize(32 * ns);
    const uint8_t* B = dev_.basepoint();
    host_parallel((ns + 7) / 8, threads_, [&](size_t g) {              // eight challenges c = H(txid, X, R) at a time
      TxStatement* tp[8];
      const uint8_t* ap[8];
      const size_t first = 8 * g, cnt = std::min<size_t>(8, ns - first);
      for (size_t q = 0; q < cnt; ++q) { tp[q] = &statement(live_all_[sg.keyed[first + q]]); ap[q] = &agg_[32 * sg.keyed[first + q]]; }
      tx_finish_signature_many(tp, ap, B, cnt);
      for (size_t q = first; q < first + cnt; ++q) {
        const TxStatement& t = *tp[q - first];
        memcpy(&sg.sst[32 * q], t.sig_scalars.data(), 32);
        memcpy(sg.ssc.data() + 32 * sg.soff[q], t.sig_scalars.data() + 32, t.sig_scalars.size() - 32);
        memcpy(sg.spt.data() + 32 * sg.soff[q], t.sig_points.data() + 32, t.sig_points.size() - 32);
      }
    });
  }
  void news() { hcv_.notify_all(); if (on_news_) on_news_(); }     // (hm_ held)
  bool keys_back(const Span& sp) const { for (size_t c = sp.first; c < sp.last; ++c) if (!keys_arrived_[c]) return false; return true; }
  // the signature stages that are due: their chunks' transaction IDs made, their keys back (hm_ held on entry and on return)
  void make_sig_stages(std::unique_lock<std::mutex>& lk, size_t ids_upto) {
    while (!quit_ && sig_next_ < sig_plan_.size() && sig_plan_[sig_next_].last <= ids_upto && keys_back(sig_plan_[sig_next_])) {
      std::unique_ptr<SigStage> sg(new SigStage());
      sg->first = sig_plan_[sig_next_].first; sg->last = sig_plan_[sig_next_].last;
      ++sig_next_;
      lk.unlock();
      const double t0 = now();
      sig_rows(*sg);
      const double dt = now() - t0;
      lk.lock();
      t_sig_host_ += dt;
      sig_stages_.push_back(std::move(sg));              // (reserved: never reallocates)
      ++n_sig_stages_;
      if (sig_next_ == sig_plan_.size()) all_sigs_made_ = true;
      news();
    }
  }
  void staging_main() {
    for (const Span& run : runs_) {                     // a run of chunks: keys and proofs of each on their way, then their IDs
      for (size_t ci = run.first; ci < run.last; ++ci) {
        Chunk& k = *chunks_[ci];
        Segment& sg = seg_[ci];
        // what the proofs need -- arity, commitments, proof bytes -- is known once the VM's stack machine has run; the keys
        // X_i and their MuSig coefficients a_i after the plan's MuSig jobs have; everything else (contract ids, anchors, the
        // transaction ID) only the signature transcripts wait for: the second pass below
        if (proofs_first_) {
          vm_pass(k, NO_PROTO);
          scan(k, sg);
          if (!gather(ci, k)) return;
          vm_pass(k, P_MUSIG);
          key_rows_out(ci, sg);
        } else {
          vm_pass(k, P_MUSIG);
          scan(k, sg);
          key_rows_out(ci, sg);
          if (!gather(ci, k)) return;
        }
      }
      for (size_t ci = run.first; ci < run.last; ++ci) {   // second pass, chunk by chunk
        {
          std::unique_lock<std::mutex> lk(hm_);
          make_sig_stages(lk, ci);
          if (quit_) return;
        }
        vm_pass(*chunks_[ci], ALL_PROTOS);
        std::lock_guard<std::mutex> lk(hm_);
        hashed_upto_ = ci + 1;
      }
    }
    std::unique_lock<std::mutex> lk(hm_);
    for (;;) {
      make_sig_stages(lk, hashed_upto_);
      if (quit_ || all_sigs_made_) return;
      hcv_.wait(lk, [&] { return quit_ || keys_back(sig_plan_[sig_next_]); });
    }
  }
  void stop_stager() {
    { std::lock_guard<std::mutex> lk(hm_); quit_ = true; }
    hcv_.notify_all();
    if (stager_.joinable()) stager_.join();
  }

  // ================================================ calling thread ================================================
  size_t n_sig_made_locked() { std::lock_guard<std::mutex> lk(hm_); return n_sig_stages_; }
  void keys_collect(size_t s) {
    Segment& sg = seg_[s];
    if (!sg.pending) return;
    sg.pending = false;
    const double t0 = now();
    const int rc = dev_.keys_collect((int)(s % n_slots_), sg.kok.data(), agg_.data() + 32 * sg.g_lo);
    if (rc != OK) note(rc, dev_.last_error());
    t_wait_ += now() - t0;
    for (size_t j = 0; j < sg.g_hi - sg.g_lo; ++j) key_ok_[sg.g_lo + j] = (sg.kok[j / 8] >> (j % 8)) & 1;
    mark("keys collected, segment", s);
  }
  void sigs_collect(size_t s) {
    SigStage& sg = *sig_stages_[s];                      // (s < the count read under hm_: the element is published)
    if (!sg.pending) return;
    sg.pending = false;
    const double t0 = now();
    const int rc = dev_.sigs_collect((int)(s % n_slots_), sg.bits.data());
    if (rc != OK) note(rc, dev_.last_error());
    t_wait_ += now() - t0;
    mark("signatures collected, stage", s);
  }
  void proofs_collect(Chunk& k) {
    if (k.handle) {
      if (k.started) {
        const double t0 = now();
        const int rc = dev_.proofs_finish(k.handle, k.pbits.data());
        if (rc != OK) note(rc, dev_.last_error());
        t_wait_ += now() - t0;
        mark("proofs collected, chunk at", k.lo);
      } else {
        dev_.proofs_release(k.handle);
      }
      k.handle = nullptr; k.started = false;
    }
    std::lock_guard<std::mutex> lk(hm_);
    arena_free_[k.index] = 1;
    hcv_.notify_all();
  }
  void enqueue_proofs(size_t ci) {
    Chunk& k = *chunks_[ci];
    mark("staged, chunk", ci);
    if (k.stage_rc != OK) { note(k.stage_rc, k.stage_err); return; }
    if (k.live.empty() || !k.handle) return;
    const double t0 = now();
    const int rc = dev_.proofs_start(ci % RING, k.handle);
    if (rc != OK) note(rc, dev_.last_error()); else k.started = true;
    t_stage_ += now() - t0;
    mark("proofs queued, chunk", ci);
  }
  // The calling thread never waits for one thing while another could be queued: it looks, in turn, for the key rows of a
  // chunk (its key stage goes out, before any of its proofs: what follows the keys is a chain, keys -> host transcripts ->
  // equations, and queued behind the proofs' chip-filling kernels its short kernels would wait for CUs), for a chunk the
  // staging thread has finished (its proofs go out), for aggregated keys that have arrived (the staging thread is told), and
  // for signature stages that are ready (the equations go out); with nothing to do it sleeps until the staging thread has
  // news, 50 us at most (what the device has finished is found out by asking).
 public:
  bool step() {
    if (finished_) return false;
    if (rc_ != OK) { finished_ = true; return true; }
    const size_t n_seg = seg_.size(), n_chunks = chunks_.size();
    bool progress = false, sigs_all, rows = false, st_ready = false, ring_free = true;
    size_t sig_avail;
    {
      std::lock_guard<std::mutex> lk(hm_);
      if (stager_failed_) { note(ENOMEM_, "out of host memory while staging the transactions"); finished_ = true; return true; }
      sig_avail = n_sig_stages_; sigs_all = all_sigs_made_;
      if (next_key_ < n_seg) rows = key_rows_[next_key_] != 0;
      if (next_stage_ < n_chunks) st_ready = staged_[next_stage_] != 0;
      if (next_stage_ >= RING) ring_free = arena_free_[next_stage_ - RING] != 0;
    }
    if (next_kcollect_ == n_seg && next_stage_ == n_chunks && sigs_all && next_sig_ == sig_avail) { finished_ = true; return true; }
    if (rows && next_kcollect_ + n_slots_ > next_key_) {                     // (its slot is free once segment next_key - n_slots is collected)
      Segment& sg = seg_[next_key_];
      mark("key rows ready, segment", next_key_);
      const double t0 = now();
      if (sg.g_hi > sg.g_lo) {
        const int rc = dev_.keys_enqueue((int)(next_key_ % n_slots_), sg.ksc.data(), sg.kpt.data(), sg.koff.data(), sg.g_hi - sg.g_lo);
        if (rc != OK) note(rc, dev_.last_error()); else sg.pending = true;
      }
      t_keys_ += now() - t0;
      mark("keys queued, segment", next_key_);
      ++next_key_;
      progress = true;
    }
    if (rc_ == OK && next_stage_ < n_chunks && (proofs_first_ || next_stage_ < next_key_)) {     // (a chunk's proofs after its keys, unless the call is one chunk)
      if (!ring_free) {                                   // the ring slot's last chunk: collected as soon as the device is through with it
        Chunk& old = *chunks_[next_stage_ - RING];
        if (!old.handle || !old.started || dev_.proofs_done(old.handle)) { proofs_collect(old); progress = true; }
      }
      if (st_ready) { enqueue_proofs(next_stage_++); progress = true; }
    }
    if (rc_ == OK && next_kcollect_ < next_key_ && (!seg_[next_kcollect_].pending || dev_.keys_done((int)(next_kcollect_ % n_slots_)))) {
      keys_collect(next_kcollect_);
      { std::lock_guard<std::mutex> lk(hm_); keys_arrived_[next_kcollect_] = 1; }
      hcv_.notify_all();
      ++next_kcollect_;
      progress = true;
    }
    while (rc_ == OK && next_sig_ < sig_avail) {
      const double t0 = now();
      if (next_sig_ >= n_slots_) {                                            // the stage that used this slot last: collected once it is done
        if (sig_stages_[next_sig_ - n_slots_]->pending && !dev_.sigs_done((int)(next_sig_ % n_slots_))) break;
        sigs_collect(next_sig_ - n_slots_);
      }
      SigStage& sg = *sig_stages_[next_sig_];
      const size_t ns = sg.keyed.size();
      if (ns) {
        const int rc = dev_.sigs_enqueue((int)(next_sig_ % n_slots_), ns, sg.ssc.data(), sg.spt.data(), sg.soff.data(), sg.sst.data());
        if (rc != OK) note(rc, dev_.last_error()); else sg.pending = true;
      }
      t_sigs_ += now() - t0;
      mark("signatures queued, stage", next_sig_);
      ++next_sig_;
      progress = true;
    }
    if (rc_ != OK) { finished_ = true; return true; }
    if (timing_) device_marks();
    return progress;
  }
 private:
  // (timing only) when the device is through with a chunk's proofs / a signature stage, as seen by the polling thread
  void device_marks() {
    if (seen_pdone_.size() != chunks_.size()) seen_pdone_.assign(chunks_.size(), 0);
    for (size_t c = 0; c < chunks_.size(); ++c) {
      Chunk& k = *chunks_[c];
      if (!seen_pdone_[c] && k.handle && k.started && dev_.proofs_done(k.handle)) { seen_pdone_[c] = 1; mark("proofs done on the device, chunk", c); }
    }
  }
  void verdicts() {
    std::vector<uint8_t> sig_ok(live_all_.size(), 0);
    for (const auto& sg : sig_stages_)
      for (size_t q = 0; q < sg->keyed.size(); ++q) if ((sg->bits[q / 8] >> (q % 8)) & 1) sig_ok[sg->keyed[q]] = 1;
    for (const auto& kp : chunks_) {
      const Chunk& k = *kp;
      for (size_t j = 0; j < k.live.size(); ++j) {
        if (!sig_ok[k.g0 + j] || !((k.pbits[j / 8] >> (j % 8)) & 1)) continue;
        const size_t i = k.lo + k.live[j];
        accept_[i / 8] |= (uint8_t)(1u << (i % 8));
        if (status_) status_[i] = TX_OK;
      }
    }
  }
  void report() const {
    if (!timing_) return;
    fprintf(stderr, "tx verify: %zu transactions in %zu chunks: staging thread: keys pass %.2f ms, VM + ids %.2f ms, gather %.2f ms, signature "
                    "transcripts %.2f ms (%zu stages); calling thread: idle %.2f ms, queueing keys %.2f, proofs %.2f, signatures %.2f ms, "
                    "waiting for the device %.2f ms; %.2f ms in all\n",
            batch_, chunks_.size(), t_keys_host_ * 1e3, t_vm_ * 1e3, t_stage_host_ * 1e3, t_sig_host_ * 1e3, sig_stages_.size(), t_wait_host_ * 1e3,
            t_keys_ * 1e3, t_stage_ * 1e3, t_sigs_ * 1e3, t_wait_ * 1e3, (now() - t00_) * 1e3);
  }

  std::vector<uint8_t> seen_pdone_;
  TxDevice& dev_;
  std::vector<TxStatement>& store_;
  std::vector<TxStatement> beyond_;
  const size_t batch_;
  std::vector<const uint8_t*> ptr_;                      // where every transaction of the call lies, and how long it is
  std::vector<size_t> len_;
  const int threads_;
  uint8_t* const accept_;
  uint8_t* const status_;
  const bool timing_;
  const double t00_;
  const size_t n_slots_;                                 // key / signature stages in flight at once (2; 1 when two calls share the stage contexts)
  std::function<void()> on_news_;
  // the plan
  std::vector<std::unique_ptr<Chunk>> chunks_;
  std::vector<Segment> seg_;                             // one key stage per chunk
  std::vector<Span> runs_, sig_plan_;                    // runs of chunks (the staging thread's unit of work); the signature stages
  bool proofs_first_ = false;
  // ---- what the two threads share (under hm_).  key_rows_[c]: the rows of chunk c are made; keys_arrived_[c]: the calling
  // thread has its encodings back; staged_[c]: chunk c has been through the VM and its statements are staged in ring slot
  // c % RING; arena_free_[c]: its proofs have been collected (the slot may be reused); sig_stages_[0 .. n_sig_stages_): the
  // signature stages made so far; all_sigs_made_; quit_; stager_failed_
  std::mutex hm_;
  std::condition_variable hcv_;
  std::vector<char> staged_, arena_free_, key_rows_, keys_arrived_;
  std::vector<std::unique_ptr<SigStage>> sig_stages_;
  size_t n_sig_stages_ = 0, sig_next_ = 0, hashed_upto_ = 0;
  bool all_sigs_made_ = false, quit_ = false, stager_failed_ = false;
  double t_keys_host_ = 0, t_sig_host_ = 0, t_stage_host_ = 0, t_vm_ = 0;
  // written by the staging thread BEFORE the flag that publishes them (key_rows_ / staged_ / n_sig_stages_), read by the
  // calling thread after it: live_all_, agg_ (the calling thread writes a segment's part before keys_arrived_), key_ok_
  std::vector<size_t> live_all_;
  std::vector<uint8_t> agg_, key_ok_;
  std::thread stager_;
  // calling thread only
  size_t next_key_ = 0, next_kcollect_ = 0, next_stage_ = 0, next_sig_ = 0;
  bool finished_ = false;
  int rc_ = OK;
  std::string error_;
  double t_keys_ = 0, t_stage_ = 0, t_sigs_ = 0, t_wait_ = 0, t_wait_host_ = 0;
};

}  // namespace zkvm
}  // namespace zk
