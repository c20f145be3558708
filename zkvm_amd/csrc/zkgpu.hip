// zkgpu.hip -- context, workspace, launch sequence and the C ABI (include/zkgpu.h).
//
// There is deliberately no CPU fallback in this file: every entry point needs a
// HIP device, and fails with ZKGPU_ENODEVICE / ZKGPU_EHIP otherwise.  The only
// arithmetic done on the host is the 255-doubling Horner tail of a *single*
// MSM (a strictly serial chain that one lane of a GPU runs ~10x slower than a
// host core) and the final RFC 9496 ENCODE of its result.
#include "../../include/zkgpu.h"
#include "../../include/zkgpu_hooks.h"   // (not exported: reached through zkgpu_hook)
#include "kernels.hpp"
#include "keccak.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <map>
#include <atomic>
#include <dirent.h>
#include <sys/random.h>
#include <unistd.h>
#include <thread>
#include <type_traits>
#include <vector>

#include "fault_gate.hpp"   // (every HIP runtime call below passes a gate a test can close: zkgpu_debug_fail_after)

#include "r1cs_verifier.hpp"
#include "cloak_plan.hpp"
#include "prep_kernels.hpp"
#include "r1cs_prover.hpp"
#include "ipa_kernels.hpp"
#include "prover_plan.hpp"
#include "prover_kernels.hpp"
#include "zkvm_tx.hpp"
#include "host_pool.hpp"
#include "ticket_cut.hpp"

using namespace zk;

namespace {

struct ProfEntry {
  const char* name;
  uint64_t launches;
  double ms;
};

struct Buffer {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct zkgpu_pointset {
  zkgpu_ctx* ctx;
  uint32_t* rows;   // n x 32 words, device
  size_t n;
  // optional fixed-base window tables (see kernels.hpp "Fixed-base path")
  uint32_t* table = nullptr;   // W x n x H niels rows
  int tbl_w = 0, tbl_W = 0;
  uint32_t tbl_H = 0;
};

struct zkgpu_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;        // generator (fixed-base) work runs beside the proof-point pipeline
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // pipelined submit / wait (see pipe_enqueue): a light stream of this context's own for the
  // latency-bound kernels; `stream` / `stream2` carry the chip-filling ones and may be SHARED with
  // the contexts forked from this one (zkgpu_ctx_fork), so that those run first-in first-out
  hipStream_t stream_l = nullptr;
  hipStream_t stream3 = nullptr;        // shared like stream/stream2: the scalar preparation (k_prepare)
  // a parent keeps STREAM_SETS (2) sets of the three shared streams; fork i uses set i mod 2:
  // consecutive batches alternate between the sets, so the latency-bound chains of two neighbours
  // overlap while each set still runs its batches first-in first-out
  std::vector<hipStream_t> lane_streams;   // owned by the parent: sets 1.. (set 0 = stream, stream2, stream3)
  int n_forks = 0;                      // live forks (each holds a hardware queue)
  unsigned fork_seq = 0;                // forks ever made: consecutive ones alternate between the stream sets
  zkgpu_ctx* parent = nullptr;
  bool owns_streams = true;
  hipEvent_t ev_t = nullptr, ev_p = nullptr, ev_sm = nullptr, ev_sa = nullptr, ev_done = nullptr;
  bool pending = false;            // a submitted batch has not been waited for yet
  size_t pending_batch = 0;
  int last_prover_slices = 0;      // slices the last prover call ran in (what the library did, not what a caller computes: ADVICE r05)
  bool broken = false;             // a device fault could not even be drained (quiesce_after_fault): every later call fails
  // the general (non-pipelined) batch paths in two halves -- batch_device_enqueue / batch_device_tables_enqueue queue the
  // kernels and the copy of the results to the pinned buffer, batch_collect waits and reads them -- so that the key and
  // signature stages of zkgpu_tx_verify_batch can run beside the host's work on the next chunk (session.hpp)
  struct SplitOp { int kind = 0; size_t batch = 0; bool values = false; hipStream_t stream = nullptr; } split;
  int pair_overlaps = -1;          // root contexts: do stream and stream2 run side by side (streams_overlap at creation; -1 not probed)
  hipEvent_t dep_event = nullptr;  // the next whole-proof submit on this context waits for it first (its inputs are still being copied)
  bool reserve_only = false;       // zkgpu_verifier_reserve: the next whole-proof enqueue sizes the workspace and launches nothing
  // A whole-proof batch queued in three pieces (session.hpp, ticket_dispatch: several device batches leaving together).  FRONT =
  // the light kernels that need the proof bytes alone -- unpack, the transcript replay; MID = the proof points' decoding and
  // tables (chip-filling); BACK = everything that waits for the transcript.  The verifier queues the FRONT of every batch,
  // then the MID of every batch, then the BACKs, so that the second batch's transcript and decoding run beside the first's
  // instead of 0.6 ms later, when the host has finished queueing the first batch's ~45 launches
  // (profiles/archive/r04am_timeline.txt).  The same call is made once per piece with the same arguments.
  enum { ENQ_ALL = 0, ENQ_FRONT = 1, ENQ_MID = 2, ENQ_BACK = 3 };
  int enqueue_phase = ENQ_ALL;
  bool awaiting_back = false;      // an ENQ_FRONT call really queued a front half (shapes outside the pipeline run whole, at once)
  std::vector<uint8_t> sync_result; // result of a submit that had to run synchronously
  bool sync_result_valid = false;
  int sync_rc = 0;
  std::recursive_mutex mu;
  std::string last_error;
  // workspace (grown on demand, never shrunk; no allocation in steady state)
  Buffer in_scalars, in_points, in_offsets, in_st_scalars, in_st_index, in_st_offsets;
  Buffer dyn_rows, bins, block_sums, entries, buckets, partials, partial_flags, window_sums, window_flags;
  Buffer msm_fail, status, accept, bitmap, ok_bytes, values, uniform;
  Buffer digits, st_partials, dynsum, accept2, bin_order, class_count, part_hist, part_entries, part_lo, dec_scratch, heavy;
  Buffer small_tbl, recoded;
  Buffer grp_sc, grp_digits, grp_partials, grp_ok, row_map;
  Buffer grp_ws, grp_wf, grp_dyn;             // window sums of the groups, their flags, the groups' proof-point sums
  Buffer grp_fail, grp_fail_sum, rechk_pts;   // failed groups: list | located candidate, their sums S1; sums of the re-checked transactions
  // the batch in flight, kept for the (never expected) ungrouped re-run when a located transaction does not explain its group
  struct LastBatch { bool valid = false, has_prep = false; const zkgpu_pointset* ps = nullptr; std::vector<uint8_t> job, prep; } last;
  bool force_unresolved = false;   // test hook: take that re-run path
  uint64_t regroup_fallbacks = 0;
  int group_size = 16;             // transactions per group check (1 = every transaction on its own)
  bool serial = false;             // measurement aid: the whole DAG of a batch on one stream
  int horner_mode = 0;             // 0 automatic, 1 one chain per transaction, 2 one per group (+ the failed groups' transactions)
  std::atomic<uint32_t> last_failed_groups{0};   // root context: failed groups in the batch finished last (any fork)
  hipEvent_t ev_dig = nullptr, ev_u = nullptr;
  Buffer prep_com, prep_proofs, prep_r, prep_pw, prep_ch, prep_wf, prep_dyn_sc, prep_dyn_pt, prep_st_sc;
  Buffer coal_com, coal_proofs, coal_r;  // merged inputs of the batches a zkgpu_verifier runs as one (session.hpp, tickets)
  Buffer ipa_lv, ipa_rv, ipa_cg, ipa_ch, ipa_w, ipa_u;   // prover: the inner-product argument's vectors (ipa_kernels.hpp)
  // the device prover (prover_kernels.hpp): constant tables, per-proof state, inputs, the rows of the phases' multiscalar
  // multiplications with their offsets / generator indices, the points coming back, the proofs
  Buffer pv_plan, pv_state, pv_in, pv_rows0, pv_rows1, pv_rows2, pv_rows3, pv_lay, pv_pts, pv_com, pv_ab, pv_proofs;
  int prover_mode = 0;             // 0: everything between the multiplications on the device, 1: host threads in lockstep
  int prover_slices = 0;           // 0: by the size of the call (prover_slice_count); test hook: a fixed number of slices
  std::vector<zkgpu_ctx*> pv_slices;    // helper contexts (one stream each) on which the slices 1.. of a prover call run (run_sliced)
  std::vector<uint32_t> pv_plan_host;   // the tables pv_plan holds (compared before uploading again)
  size_t pv_lay_batch = 0;              // batch size the scaffolding in pv_lay was built for (0: none)
  Buffer prep_absorb, prep_raw;    // cooperative transcript: absorbed words per segment, raw challenge bytes
  int locate_mode = 0;             // failed groups: 0 automatic, 1 always re-check every transaction, 2 always locate the culprit
  int transcript_mode = 0;         // 0 automatic, 1 one lane per transaction, 2 one wavefront per transaction
  int forced_parts = 0;
  int locate_parts = 0;            // lanes per (failed group, window) of the locating multiplication (0: 32)
  int tail_mode = 0;               // 0: the tail's sums inside k_locate_fused / k_recheck_fused; 1: launches of their own
  void* pinned = nullptr;   // host staging for results
  size_t pinned_cap = 0;
  void* pinned_in = nullptr;   // host staging for inputs handed over in host memory
  size_t pinned_in_cap = 0;
  // profiling
  bool profiling = false;
  std::vector<ProfEntry> prof;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  std::vector<std::pair<int, int>> ev_used;   // (prof index, pool index)
  size_t ev_next = 0;
  int forced_w = 0;
  int last_w = 0;
  uint64_t last_adds = 0;
};

namespace {
// A runtime call of a context's pipeline has failed.  What the pipeline had queued BEFORE it is still in flight -- on the
// context's own streams and on the streams it shares with its family -- while the caller is about to treat the context as
// idle: a lane of the verifier would be handed its next batch, whose first kernels (light stream) overwrite the workspace the
// failed batch's last kernels (shared streams) still read; a staging area would be given to the next tickets.  Found by fault
// injection (tests/test_gpu_faults.py: a failed wait on one device batch flipped a bit of the NEXT batch on that lane, status
// OK).  So the error does not leave the library before the device is drained; if even that fails the context is broken for
// good and every later call on it fails (an error is never an accept).  The sticky launch error is read off so that it is not
// blamed on the next batch.  Error path only; the gate of fault_gate.hpp is held open (the fault under test has happened).
void quiesce_after_fault(zkgpu_ctx* c) {
  zk::fault::Suppress open;
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != c->device) (void)hipSetDevice(c->device);
  if (hipDeviceSynchronize() != hipSuccess) c->broken = true;
  (void)hipGetLastError();
  if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
}

#define HIP_TRY(ctx, expr)                                                                  \
  do {                                                                                      \
    if ((ctx)->broken) {                                                                    \
      (ctx)->last_error = "an earlier device fault on this context could not be drained: the context is unusable"; \
      return ZKGPU_EHIP;                                                                    \
    }                                                                                       \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      char buf__[256];                                                                      \
      snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
               __FILE__, __LINE__);                                                         \
      (ctx)->last_error = buf__;                                                            \
      quiesce_after_fault(ctx);                                                             \
      return ZKGPU_EHIP;                                                                    \
    }                                                                                       \
  } while (0)

int ensure(zkgpu_ctx* c, Buffer& b, size_t bytes) {
  if (bytes <= b.cap) return ZKGPU_OK;
  if (b.p) HIP_TRY(c, hipFree(b.p));
  b.p = nullptr; b.cap = 0;
  size_t want = bytes + bytes / 8 + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { c->last_error = std::string("hipMalloc: ") + hipGetErrorString(e); b.p = nullptr; quiesce_after_fault(c); return ZKGPU_ENOMEM; }
  b.cap = want;
  return ZKGPU_OK;
}

int ensure_pinned(zkgpu_ctx* c, size_t bytes) {
  if (bytes <= c->pinned_cap) return ZKGPU_OK;
  if (c->pinned) HIP_TRY(c, hipHostFree(c->pinned));
  c->pinned = nullptr; c->pinned_cap = 0;
  HIP_TRY(c, hipHostMalloc(&c->pinned, bytes + 4096, hipHostMallocDefault));
  c->pinned_cap = bytes + 4096;
  return ZKGPU_OK;
}

#define TRY(expr) do { int rc__ = (expr); if (rc__ != ZKGPU_OK) return rc__; } while (0)

// A context with a submitted batch still to be waited for owns its workspace, its status words and its pinned result
// buffer on behalf of that batch: every synchronous entry point refuses to run on it (it would overwrite them, and the
// batch's verdicts with them) until zkgpu_verify_wait has collected the batch.  Call with c->mu held.
int refuse_if_pending(zkgpu_ctx* c) {
  if (!c->pending) return ZKGPU_OK;
  c->last_error = "a submitted batch is still waiting for zkgpu_verify_wait on this context";
  return ZKGPU_EINVAL;
}

// Verifier randomness (the weight r of each proof's two equation halves and of a transaction inside
// a group check is the soundness parameter): from the kernel's CSPRNG, getrandom(2).  false = no
// randomness available: the caller fails closed.
bool os_random(void* out, size_t n) {
  uint8_t* p = (uint8_t*)out;
  while (n) {
    const ssize_t got = getrandom(p, n, 0);
    if (got < 0) { if (errno == EINTR) continue; return false; }
    p += got; n -= (size_t)got;
  }
  return true;
}

int prof_index(zkgpu_ctx* c, const char* name) {
  for (size_t i = 0; i < c->prof.size(); ++i) if (strcmp(c->prof[i].name, name) == 0) return (int)i;
  c->prof.push_back({name, 0, 0.0});
  return (int)c->prof.size() - 1;
}

struct Launch {
  zkgpu_ctx* c;
  hipStream_t st;
  int pool = -1;
  Launch(zkgpu_ctx* ctx, const char* name, hipStream_t stream = nullptr) : c(ctx), st(stream ? stream : ctx->stream) {
    if (!c->profiling) return;
    if (c->ev_next == c->ev_pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      c->ev_pool.push_back({a, b});
    }
    pool = (int)c->ev_next++;
    c->ev_used.push_back({prof_index(c, name), pool});
    (void)hipEventRecord(c->ev_pool[pool].first, st);
  }
  ~Launch() {
    if (pool >= 0) (void)hipEventRecord(c->ev_pool[pool].second, st);
  }
};

// ZKGPU_TIMELINE=<file>: every profiled launch as "ctx name start_ms end_ms" on one device-wide clock
// (a debugging aid for the batches-in-flight pipeline; the reference event is recorded at first use)
hipEvent_t g_timeline_ref = nullptr;
FILE* g_timeline = nullptr;
std::mutex g_timeline_mu;

void prof_collect(zkgpu_ctx* c) {
  static const char* tl_path = getenv("ZKGPU_TIMELINE");
  for (auto& u : c->ev_used) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev_pool[u.second].first, c->ev_pool[u.second].second) == hipSuccess) {
      c->prof[u.first].launches += 1;
      c->prof[u.first].ms += ms;
    }
    if (tl_path && g_timeline_ref) {
      std::lock_guard<std::mutex> lk(g_timeline_mu);
      if (!g_timeline) g_timeline = fopen(tl_path, "w");
      float a = 0, b = 0;
      if (g_timeline && hipEventElapsedTime(&a, g_timeline_ref, c->ev_pool[u.second].first) == hipSuccess &&
          hipEventElapsedTime(&b, g_timeline_ref, c->ev_pool[u.second].second) == hipSuccess) {
        fprintf(g_timeline, "%p %s %.4f %.4f\n", (void*)c, c->prof[u.first].name, a, b);
        fflush(g_timeline);
      }
    }
  }
  c->ev_used.clear();
  c->ev_next = 0;
}

// wavefronts per MSM in k_small_accumulate: about three per SIMD (1024 SIMDs) over the whole launch
inline int small_parts(uint64_t B) {
  return (int)std::max<uint64_t>(1, std::min<uint64_t>(4, (3072 + B / 2) / std::max<uint64_t>(B, 1)));
}

inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// sets of shared chip-filling streams per parent context (consecutive forks alternate between them) and the
// fork limit that keeps the process under the ~22 hardware queues the runtime hands out before it
// multiplexes them in software (measured: 100-200 ms per step beyond that)
constexpr int TAIL_THREADS = 512;   // threads per row in k_recheck_fused (shares of the row's generator sum)
constexpr int LOCATE_THREADS = 512; // ... in k_locate_fused: one wavefront per SIMD, so that the naming of the culprit has 512 registers and no scratch
constexpr int STREAM_SETS = 2;
constexpr int MAX_FORKS = 9;
constexpr size_t LOCATE_MIN_BATCH = 2048;      // transactions per batch from which failed groups are located instead of re-checked in full
constexpr size_t COOP_TRANSCRIPT_MAX = 1536;   // transactions per batch up to which the transcript runs one wavefront each

// window width minimising  W * (terms + 2 * 2^(w-1) * msms)  point additions
int choose_window(uint64_t n_terms, uint32_t n_msm) {
  double per = (double)n_terms / (n_msm ? n_msm : 1);
  int best = 4; double best_cost = 1e300;
  for (int w = 4; w <= 16; ++w) {
    double W = 255 / w + 1;
    double reduce = 2.0 * (double)(1u << (w - 1));
    if ((1u << (w - 1)) > (unsigned)REDUCE_CHUNK) reduce *= 1.3;  // chunk fix-up work
    double cost = W * (per + reduce);
    if (cost < best_cost) { best_cost = cost; best = w; }
  }
  return best;
}

struct Job {
  // device pointers
  const uint32_t* d_dyn_scalars = nullptr;
  const uint32_t* d_dyn_points = nullptr;
  const uint64_t* d_dyn_offsets = nullptr;
  uint64_t n_dyn = 0;
  const uint32_t* d_st_scalars = nullptr;
  const uint32_t* d_st_index = nullptr;
  const uint64_t* d_st_offsets = nullptr;
  uint64_t n_static = 0;
  const uint32_t* d_static_rows = nullptr;
  uint32_t n_msm = 1;
  const uint32_t* d_wellformed = nullptr;   // optional per-MSM flags ANDed into the accept bitmap
  uint64_t max_dyn_row = ~0ull;             // longest row of dynamic terms, when the caller's offsets are in host memory (~0: not known)
};
inline uint64_t longest_row(const uint64_t* offsets, size_t batch) {
  uint64_t m = 0;
  for (size_t i = 0; i < batch; ++i) m = std::max(m, offsets[i + 1] - offsets[i]);
  return m;
}

// Runs decompress .. window sums.  On return (stream not yet synchronised):
//   c->window_sums / c->window_flags hold n_msm * n_windows extended points
//   c->msm_fail[m] != 0 when MSM m had an undecodable point
//   c->status: [0] flags (bit 1: scalar >= 2^255), [2..3] u64 min bad point index
int run_to_windows(zkgpu_ctx* c, const Job& job, JobDesc& jd, bool reset_status = true) {
  const uint64_t n_terms = job.n_dyn + job.n_static;
  int w = c->forced_w ? c->forced_w : choose_window(n_terms, job.n_msm);
  w = std::max(2, std::min(16, w));
  jd.dyn_scalars = job.d_dyn_scalars;
  jd.dyn_offsets = job.d_dyn_offsets;
  jd.n_dyn = job.n_dyn;
  jd.st_scalars = job.d_st_scalars;
  jd.st_index = job.d_st_index;
  jd.st_offsets = job.d_st_offsets;
  jd.n_static = job.n_static;
  jd.n_msm = job.n_msm;
  jd.w = w;
  jd.n_windows = 255 / w + 1;
  jd.n_buckets = 1u << (w - 1);
  c->last_w = w;

  const uint64_t n_wins = (uint64_t)job.n_msm * jd.n_windows;
  const uint64_t n_bins = n_wins * jd.n_buckets;
  const uint32_t chunk_size = jd.n_buckets <= (uint32_t)REDUCE_CHUNK ? jd.n_buckets : (uint32_t)REDUCE_CHUNK_BIG;
  const uint32_t chunks = (jd.n_buckets + chunk_size - 1) / chunk_size;
  const uint64_t n_tasks = n_wins * chunks;
  const uint64_t max_entries = n_terms * jd.n_windows;
  if (max_entries >= (1ull << 32) || n_bins >= (1ull << 32) || job.n_dyn >= (1ull << 30) ||
      job.n_static >= (1ull << 32)) {
    c->last_error = "job too large for 32-bit entry indices; split the batch";
    return ZKGPU_EINVAL;
  }
  const unsigned scan_blocks = blocks_for(n_bins, SCAN_TILE);

  TRY(ensure(c, c->dyn_rows, std::max<uint64_t>(job.n_dyn, 1) * NIELS_WORDS * 4));
  TRY(ensure(c, c->bins, (n_bins + 1) * 4));
  TRY(ensure(c, c->block_sums, ((size_t)scan_blocks + 1) * 4));
  TRY(ensure(c, c->entries, std::max<uint64_t>(max_entries, 1) * 4));
  TRY(ensure(c, c->buckets, n_bins * EXT_WORDS * 4));
  TRY(ensure(c, c->partials, n_tasks * EXT_WORDS * 4));
  TRY(ensure(c, c->partial_flags, n_tasks * 4));
  TRY(ensure(c, c->window_sums, n_wins * EXT_WORDS * 4));
  TRY(ensure(c, c->window_flags, n_wins * 4));
  TRY(ensure(c, c->msm_fail, (size_t)job.n_msm * 4));
  TRY(ensure(c, c->status, 64));

  // heavy list | fat list (kernels.hpp: bins summed by a workgroup each / by FAT_LANES lanes each)
  const uint32_t fat_cap = (uint32_t)(max_entries / FAT_BIN + 64);
  TRY(ensure(c, c->heavy, ((size_t)HEAVY_MAX + 1 + fat_cap + 1) * 4));
  uint32_t* heavy = (uint32_t*)c->heavy.p;
  uint32_t* fat = heavy + HEAVY_MAX + 1;
  TRY(ensure(c, c->class_count, 2 * SIZE_CLASSES * 4));

  hipStream_t s = c->stream;
  // (every counter the sort and the bin order start from is cleared HERE, before anything runs: a memset between two kernels is
  // a launch of its own, ~10 us of an otherwise empty queue each -- four of them stood between the sort and the accumulation)
  HIP_TRY(c, hipMemsetAsync(heavy, 0, 4, s));
  HIP_TRY(c, hipMemsetAsync(fat, 0, 4, s));
  HIP_TRY(c, hipMemsetAsync(c->class_count.p, 0, SIZE_CLASSES * 4, s));
  const bool part_sort = job.n_msm == 1 && job.n_static == 0 && w - 1 >= PART_LO_BITS && n_terms >= 32768 && n_terms <= (1ull << PART_IDX_BITS);
  if (!part_sort) HIP_TRY(c, hipMemsetAsync(c->bins.p, 0, (n_bins + 1) * 4, s));     // (the partition sort writes every bin's end offset itself)
  HIP_TRY(c, hipMemsetAsync(c->msm_fail.p, 0, (size_t)job.n_msm * 4, s));
  if (reset_status) {
    HIP_TRY(c, hipMemsetAsync(c->status.p, 0, 8, s));
    HIP_TRY(c, hipMemsetAsync((char*)c->status.p + 8, 0xff, 8, s));
  }
  uint32_t* status = (uint32_t*)c->status.p;
  unsigned long long* bad_index = (unsigned long long*)((char*)c->status.p + 8);

  bool decompress_aside = false;
  if (job.n_dyn >= 131072) {
    // many points: run the squaring chain in its own lean kernel (kernels.hpp "split decompression"),
    // and on the second stream: the digit sort below needs the scalars only, is bound by LDS atomics
    // and memory while this is pure integer VALU work, so the two overlap; joined before the buckets
    TRY(ensure(c, c->dec_scratch, job.n_dyn * DEC_WORDS * 4));
    hipStream_t sd = c->stream2 ? c->stream2 : s;
    decompress_aside = sd != s;
    if (decompress_aside) {
      HIP_TRY(c, hipEventRecord(c->ev_fork, s));
      HIP_TRY(c, hipStreamWaitEvent(sd, c->ev_fork, 0));
    }
    {
      Launch l(c, "k_decompress", sd);
      hipLaunchKernelGGL(k_decompress_pre, dim3(blocks_for(job.n_dyn, 256)), dim3(256), 0, sd, job.d_dyn_points,
                         (uint32_t*)c->dec_scratch.p, job.n_dyn);
      hipLaunchKernelGGL(k_pow22523, dim3(blocks_for(job.n_dyn, 256)), dim3(256), 0, sd, (uint32_t*)c->dec_scratch.p,
                         job.n_dyn);
      hipLaunchKernelGGL(k_decompress_post, dim3(blocks_for(job.n_dyn, 256)), dim3(256), 0, sd, job.d_dyn_points,
                         (const uint32_t*)c->dec_scratch.p, (uint32_t*)c->dyn_rows.p, job.n_dyn, job.d_dyn_offsets,
                         job.n_msm, (uint32_t*)c->msm_fail.p, bad_index);
    }
    if (decompress_aside) HIP_TRY(c, hipEventRecord(c->ev_join, sd));
  } else if (job.n_dyn) {
    Launch l(c, "k_decompress");
    hipLaunchKernelGGL(k_decompress, dim3(blocks_for(job.n_dyn, 256)), dim3(256), 0, s, job.d_dyn_points,
                       (uint32_t*)c->dyn_rows.p, job.n_dyn, job.d_dyn_offsets, job.n_msm,
                       (uint32_t*)c->msm_fail.p, bad_index, (uint8_t*)nullptr);
  }
  uint32_t* class_count = (uint32_t*)c->class_count.p;
  uint32_t* class_cursor = class_count + SIZE_CLASSES;
  if (part_sort) {
    // single large MSM: two-level sort with LDS atomics only
    PartShape ps;
    ps.lo_bits = PART_LO_BITS;
    ps.hi_bits = (uint32_t)(w - 1 - PART_LO_BITS);
    ps.n_part = (uint32_t)jd.n_windows << ps.hi_bits;
    ps.n_tiles = (uint32_t)blocks_for(n_terms, PART_TILE);
    const uint64_t n_cells = (uint64_t)ps.n_part * ps.n_tiles;
    TRY(ensure(c, c->part_hist, n_cells * 4));
    TRY(ensure(c, c->block_sums, ((size_t)2 * ps.n_part + 4) * 4));        // totals[n_part] | part_base[n_part + 1] | ticket
    TRY(ensure(c, c->part_entries, std::max<uint64_t>(max_entries, 1) * 4));
    uint32_t* totals = (uint32_t*)c->block_sums.p;
    uint32_t* part_base = totals + ps.n_part;
    uint32_t* ticket = part_base + ps.n_part + 1;
    HIP_TRY(c, hipMemsetAsync(ticket, 0, 4, s));
    {
      Launch l(c, "k_part_hist");
      hipLaunchKernelGGL(k_part_hist, dim3(ps.n_tiles), dim3(256), ps.n_part * 4, s, jd, ps, (uint32_t*)c->part_hist.p,
                         status);
    }
    {
      Launch l(c, "k_part_offsets");
      hipLaunchKernelGGL(k_part_offsets, dim3(ps.n_part), dim3(256), 0, s, ps, (uint32_t*)c->part_hist.p, totals, part_base, ticket);
    }
    {
      Launch l(c, "k_part_scatter");
      hipLaunchKernelGGL(k_part_scatter, dim3(ps.n_tiles), dim3(256), ps.n_part * 4, s, jd, ps,
                         (const uint32_t*)c->part_hist.p, (const uint32_t*)part_base, (uint32_t*)c->part_entries.p);
    }
    {
      Launch l(c, "k_part_sort");
      hipLaunchKernelGGL(k_part_sort, dim3(ps.n_part), dim3(256), PART_SORT_LDS * 4, s, ps, (const uint32_t*)part_base,
                         (const uint32_t*)c->part_entries.p, (uint32_t*)c->entries.p, (uint32_t*)c->bins.p, class_count);
    }
  } else {
    if (n_terms) {
      Launch l(c, "k_digits_count");
      hipLaunchKernelGGL(k_digits_count, dim3(blocks_for(n_terms, 256)), dim3(256), 0, s, jd, (uint32_t*)c->bins.p,
                         status);
    }
    {
      Launch l(c, "k_scan");
      hipLaunchKernelGGL(k_scan_reduce, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, s, (const uint32_t*)c->bins.p, n_bins,
                         (uint32_t*)c->block_sums.p);
      hipLaunchKernelGGL(k_scan_blocksums, dim3(1), dim3(SCAN_BLOCK), 0, s, (uint32_t*)c->block_sums.p, scan_blocks);
      hipLaunchKernelGGL(k_scan_apply, dim3(scan_blocks), dim3(SCAN_BLOCK), 0, s, (uint32_t*)c->bins.p, n_bins,
                         (const uint32_t*)c->block_sums.p);
    }
    if (n_terms) {
      Launch l(c, "k_digits_scatter");
      hipLaunchKernelGGL(k_digits_scatter, dim3(blocks_for(n_terms, 256)), dim3(256), 0, s, jd, (uint32_t*)c->bins.p,
                         (uint32_t*)c->entries.p);
    }
  }
  TRY(ensure(c, c->bin_order, n_bins * 4));
  {
    Launch l(c, "k_bin_order");
    if (!part_sort)                                      // (the partition sort counts the size classes of its bins itself)
      hipLaunchKernelGGL(k_bin_classes, dim3(blocks_for(n_bins, 256)), dim3(256), 0, s, (const uint32_t*)c->bins.p, n_bins,
                         class_count);
    hipLaunchKernelGGL(k_class_scan, dim3(1), dim3(256), 0, s, (const uint32_t*)class_count, class_cursor);
    hipLaunchKernelGGL(k_bin_order, dim3(blocks_for(n_bins, 256)), dim3(256), 0, s, (const uint32_t*)c->bins.p, n_bins,
                       class_cursor, (uint32_t*)c->bin_order.p, heavy, fat, fat_cap);
  }
  if (decompress_aside) HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
  {
    // one launch: its first workgroups sum the fat bins with FAT_LANES lanes each (they hold the longest chains and must start
    // first), the others one bin per lane
    const unsigned fat_blocks = std::min<unsigned>(blocks_for((uint64_t)fat_cap * FAT_LANES, 256), 320u);
    Launch l(c, "k_bucket_accumulate");
    hipLaunchKernelGGL(k_bucket_accumulate, dim3(fat_blocks + blocks_for(n_bins, 256)), dim3(256), 0, s,
                       (const uint32_t*)c->bins.p, (const uint32_t*)c->entries.p, job.d_static_rows,
                       (const uint32_t*)c->dyn_rows.p, (uint32_t*)c->buckets.p, n_bins,
                       (const uint32_t*)c->bin_order.p, (const uint32_t*)fat, fat_cap, fat_blocks);
  }
  {
    Launch l(c, "k_bucket_heavy");
    hipLaunchKernelGGL(k_bucket_heavy, dim3(512), dim3(256), 0, s, (const uint32_t*)c->bins.p,
                       (const uint32_t*)c->entries.p, job.d_static_rows, (const uint32_t*)c->dyn_rows.p,
                       (uint32_t*)c->buckets.p, (const uint32_t*)heavy);
  }
  uint32_t* partials = chunks == 1 ? (uint32_t*)c->window_sums.p : (uint32_t*)c->partials.p;
  uint32_t* pflags = chunks == 1 ? (uint32_t*)c->window_flags.p : (uint32_t*)c->partial_flags.p;
  {
    // few tasks (one large multiscalar multiplication: 32 768): a quad of lanes per task, a third of the chain's depth; many
    // tasks (a batch of small ones): one lane each, a quarter of the instructions
    Launch l(c, "k_bucket_reduce");
    if (n_tasks <= 65536)
      hipLaunchKernelGGL(k_bucket_reduce_quad, dim3(blocks_for(4 * n_tasks, 256)), dim3(256), 0, s, (const uint32_t*)c->bins.p,
                         (const uint32_t*)c->buckets.p, partials, pflags, n_tasks, jd.n_buckets, chunks, chunk_size);
    else
      hipLaunchKernelGGL(k_bucket_reduce, dim3(blocks_for(n_tasks, 256)), dim3(256), 0, s, (const uint32_t*)c->bins.p,
                         (const uint32_t*)c->buckets.p, partials, pflags, n_tasks, jd.n_buckets, chunks, chunk_size);
  }
  if (chunks > 1) {
    Launch l(c, "k_window_partials");
    hipLaunchKernelGGL(k_window_partials, dim3((unsigned)n_wins), dim3(256), 0, s, (const uint32_t*)c->partials.p,
                       (const uint32_t*)c->partial_flags.p, (uint32_t*)c->window_sums.p,
                       (uint32_t*)c->window_flags.p, chunks);
  }
  HIP_TRY(c, hipGetLastError());
  return ZKGPU_OK;
}

// ---- single MSM ----------------------------------------------------------------
int msm_device(zkgpu_ctx* c, const void* d_scalars, const void* d_points, size_t n, uint8_t out[32],
               size_t* bad_index) {
  memset(out, 0, 32);
  if (bad_index) *bad_index = (size_t)-1;
  if (n == 0) return ZKGPU_OK;   // empty sum = identity = all-zero encoding
  Job job;
  job.d_dyn_scalars = (const uint32_t*)d_scalars;
  job.d_dyn_points = (const uint32_t*)d_points;
  job.n_dyn = n;
  job.n_msm = 1;
  JobDesc jd;
  TRY(run_to_windows(c, job, jd));
  const size_t W = (size_t)jd.n_windows;
  const size_t bytes = W * EXT_WORDS * 4 + W * 4 + 64;
  TRY(ensure_pinned(c, bytes));
  char* h = (char*)c->pinned;
  HIP_TRY(c, hipMemcpyAsync(h, c->window_sums.p, W * EXT_WORDS * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(h + W * EXT_WORDS * 4, c->window_flags.p, W * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(h + W * EXT_WORDS * 4 + W * 4, c->status.p, 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->profiling) prof_collect(c);
  const uint32_t* hs = (const uint32_t*)(h + W * EXT_WORDS * 4 + W * 4);
  unsigned long long bad;
  memcpy(&bad, hs + 2, 8);
  if (bad != ~0ull) {
    if (bad_index) *bad_index = (size_t)bad;
    return ZKGPU_EINVALID_POINT;
  }
  if (hs[0] & 2u) { c->last_error = "scalar with bit 255 set"; return ZKGPU_EINVAL; }
  // Horner over the windows on the host (serial tail, see file header)
  const uint32_t* sums = (const uint32_t*)h;
  const uint32_t* flags = (const uint32_t*)(h + W * EXT_WORDS * 4);
  ge acc;
  bool have = false;
  for (int t = (int)W - 1; t >= 0; --t) {
    if (have) {
      for (int k = 0; k < jd.w - 1; ++k) ge_double<false>(acc, acc);
      ge_double<true>(acc, acc);
    }
    if (flags[t]) {
      ge p;
      const uint32_t* r = sums + (size_t)t * EXT_WORDS;
      for (int i = 0; i < 10; ++i) { p.X.v[i] = r[i]; p.Y.v[i] = r[10 + i]; p.Z.v[i] = r[20 + i]; p.T.v[i] = r[30 + i]; }
      if (have) ge_add(acc, acc, p); else { acc = p; have = true; }
    }
  }
  if (!have) return ZKGPU_OK;
  uint32_t enc[8];
  ristretto_encode(enc, acc);
  memcpy(out, enc, 32);
  return ZKGPU_OK;
}

// ---- batch ------------------------------------------------------------------------
// values != nullptr: "value mode" -- write the 32-byte encoding of every MSM to
// values[32 * i] (host) and make bit i mean "all points of MSM i decoded".
int small_msm_launch(zkgpu_ctx* c, const Job& job, hipStream_t st);

int batch_device_enqueue(zkgpu_ctx* c, const Job& job, bool values) {
  const size_t B = job.n_msm;
  const size_t nbytes = (B + 7) / 8;
  c->split = zkgpu_ctx::SplitOp{1, B, values, c->stream};
  if (B == 0) return ZKGPU_OK;
  JobDesc jd;
  if (!c->forced_w && job.n_static == 0 && job.n_dyn && job.n_dyn <= 64ull * B && B >= 64 && job.max_dyn_row <= 256) {
    // many SMALL multiscalar multiplications (a few terms each: aggregated keys, signature equations): per-point tables
    // and one wavefront per row -- four launches, no global sort, no atomics -- instead of the bucket pipeline's seventeen.
    // EVERY row must be short, not just the average: one workgroup walks a row serially, and the tables are 1280 B per
    // term -- a batch with one very long row among many short ones (or offsets this side cannot see: the _dev entry points)
    // takes the bucket pipeline, which spreads a row over the chip
    hipStream_t s = c->stream;
    jd.w = 4; jd.n_windows = 64;
    c->last_w = 4;
    TRY(ensure(c, c->dyn_rows, std::max<uint64_t>(job.n_dyn, 1) * NIELS_WORDS * 4));
    TRY(ensure(c, c->window_sums, (size_t)B * 64 * EXT_WORDS * 4));
    TRY(ensure(c, c->window_flags, (size_t)B * 64 * 4));
    TRY(ensure(c, c->msm_fail, (size_t)B * 4));
    TRY(ensure(c, c->status, 64));
    {
      Launch l(c, "k_batch_init", s);
      hipLaunchKernelGGL(k_batch_init, dim3(blocks_for(B, 256)), dim3(256), 0, s, (uint32_t*)c->status.p, (uint32_t*)c->msm_fail.p,
                         (uint32_t*)nullptr, (uint32_t)B);
    }
    {
      Launch l(c, "k_decompress", s);
      hipLaunchKernelGGL(k_decompress, dim3(blocks_for(job.n_dyn, 256)), dim3(256), 0, s, job.d_dyn_points,
                         (uint32_t*)c->dyn_rows.p, job.n_dyn, job.d_dyn_offsets, (uint32_t)B,
                         (uint32_t*)c->msm_fail.p, (unsigned long long*)((char*)c->status.p + 8), (uint8_t*)nullptr);
    }
    TRY(small_msm_launch(c, job, s));
  } else {
    TRY(run_to_windows(c, job, jd));
  }
  TRY(ensure(c, c->accept, B));
  TRY(ensure(c, c->bitmap, nbytes));
  TRY(ensure_pinned(c, nbytes + 64 + (values ? 32 * B : 0)));
  if (values) TRY(ensure(c, c->values, 32 * B));
  {
    Launch l(c, "k_msm_finish_quad");
    hipLaunchKernelGGL(k_msm_finish_quad, dim3(blocks_for(4 * B, 256)), dim3(256), 0, c->stream,
                       (const uint32_t*)c->window_sums.p, (const uint32_t*)c->window_flags.p,
                       (const uint32_t*)c->msm_fail.p, (uint8_t*)c->accept.p,
                       values ? (uint32_t*)c->values.p : (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t)B, jd.w,
                       jd.n_windows, (const uint32_t*)nullptr, 0u);
  }
  {
    Launch l(c, "k_pack_bitmap");
    hipLaunchKernelGGL(k_pack_bitmap, dim3(blocks_for(nbytes, 256)), dim3(256), 0, c->stream,
                       (const uint8_t*)c->accept.p, job.d_wellformed, (uint8_t*)c->bitmap.p, (uint32_t)B);
  }
  HIP_TRY(c, hipGetLastError());
  char* h = (char*)c->pinned;
  HIP_TRY(c, hipMemcpyAsync(h, c->bitmap.p, nbytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(h + nbytes, c->status.p, 16, hipMemcpyDeviceToHost, c->stream));
  if (values) HIP_TRY(c, hipMemcpyAsync(h + nbytes + 64, c->values.p, 32 * B, hipMemcpyDeviceToHost, c->stream));
  return ZKGPU_OK;
}

// second half of batch_device_enqueue / batch_device_tables_enqueue: wait for the stream, read the results
int batch_collect(zkgpu_ctx* c, uint8_t* accept_bitmap, uint8_t* values) {
  const zkgpu_ctx::SplitOp op = c->split;
  c->split = zkgpu_ctx::SplitOp{};
  const size_t B = op.batch, nbytes = (B + 7) / 8;
  memset(accept_bitmap, 0, nbytes);
  if (values) memset(values, 0, 32 * B);
  if (op.kind == 0) return ZKGPU_EINVAL;
  if (B == 0) return ZKGPU_OK;
  HIP_TRY(c, hipStreamSynchronize(op.stream));
  if (c->profiling) prof_collect(c);
  const char* h = (const char*)c->pinned;
  uint32_t st;
  memcpy(&st, h + nbytes, 4);
  if (st & 2u) { c->last_error = "scalar with bit 255 set"; return ZKGPU_EINVAL; }
  memcpy(accept_bitmap, h, nbytes);
  if (values && op.values) memcpy(values, h + nbytes + 64, 32 * B);
  return ZKGPU_OK;
}

// values != nullptr: "value mode" -- write the 32-byte encoding of every MSM to
// values[32 * i] (host) and make bit i mean "all points of MSM i decoded".
int batch_device(zkgpu_ctx* c, const Job& job, uint8_t* accept_bitmap, uint8_t* values = nullptr) {
  const size_t B = job.n_msm;
  memset(accept_bitmap, 0, (B + 7) / 8);
  if (values) memset(values, 0, 32 * B);
  const int rc = batch_device_enqueue(c, job, values != nullptr);
  if (rc != ZKGPU_OK) { c->split = zkgpu_ctx::SplitOp{}; return rc; }
  return batch_collect(c, accept_bitmap, values);
}

// window sums of many small MSMs (few proof-specific points each): see kernels.hpp, k_small_tables.
// d_scalars == nullptr: the scalars are not known yet; whoever produces them writes c->recoded too.
int small_tables_launch(zkgpu_ctx* c, const uint32_t* d_scalars, uint64_t n_dyn, hipStream_t st) {
  TRY(ensure(c, c->small_tbl, std::max<uint64_t>(n_dyn, 1) * SMALL_TBL * EXT_WORDS * 4));
  TRY(ensure(c, c->recoded, std::max<uint64_t>(n_dyn, 1) * 32));
  Launch l(c, "k_small_tables", st);
  hipLaunchKernelGGL(k_small_tables, dim3(blocks_for(n_dyn, 64)), dim3(64), 0, st, d_scalars,
                     (const uint32_t*)c->dyn_rows.p, n_dyn, (uint32_t*)c->small_tbl.p, (uint32_t*)c->recoded.p,
                     (uint32_t*)c->status.p);
  return ZKGPU_OK;
}

int small_accumulate_launch(zkgpu_ctx* c, const Job& job, hipStream_t st) {
  const size_t B = job.n_msm;
  Launch l(c, "k_small_accumulate", st);
  const int parts = small_parts(B);
  hipLaunchKernelGGL(k_small_accumulate, dim3((unsigned)B), dim3(64 * parts), (size_t)(parts - 1) * 41 * 64 * 4, st,
                     (const uint32_t*)c->recoded.p, job.d_dyn_offsets, (const uint32_t*)c->small_tbl.p, (uint32_t)B,
                     (uint32_t*)c->window_sums.p, (uint32_t*)c->window_flags.p);
  return ZKGPU_OK;
}

int small_msm_launch(zkgpu_ctx* c, const Job& job, hipStream_t st) {
  TRY(small_tables_launch(c, job.d_dyn_scalars, job.n_dyn, st));
  return small_accumulate_launch(c, job, st);
}

// Batch path when the point set carries fixed-base tables: generator terms are
// summed straight out of the tables (no sort, no buckets, no doublings); only
// the proof-specific terms go through the Pippenger pipeline.
int batch_device_tables_enqueue(zkgpu_ctx* c, const Job& job, const zkgpu_pointset* ps) {
  const size_t B = job.n_msm;
  const size_t nbytes = (B + 7) / 8;
  c->split = zkgpu_ctx::SplitOp{2, B, false, c->stream};
  if (B == 0) return ZKGPU_OK;
  hipStream_t s = c->stream, s2 = c->stream2;
  const bool has_dyn = job.n_dyn > 0;
  const int W = ps->tbl_W;
  int P = c->forced_parts;
  if (P <= 0) {
    P = (int)((131072 + B * W - 1) / (B * W));
    P = std::max(1, std::min(16, P));
  }
  const uint64_t n_lanes = (uint64_t)B * W * P;
  TRY(ensure(c, c->accept, B));
  TRY(ensure(c, c->accept2, B));
  TRY(ensure(c, c->bitmap, nbytes));
  TRY(ensure_pinned(c, nbytes + 64));
  TRY(ensure(c, c->status, 64));
  TRY(ensure(c, c->digits, std::max<uint64_t>(job.n_static, 1) * W * 2));
  TRY(ensure(c, c->st_partials, n_lanes * EXT_WORDS * 4));
  TRY(ensure(c, c->dynsum, B * EXT_WORDS * 4));
  HIP_TRY(c, hipMemsetAsync(c->status.p, 0, 8, s));
  HIP_TRY(c, hipMemsetAsync((char*)c->status.p + 8, 0xff, 8, s));
  // fork: generator terms on stream2 (fills the chip), proof-point pipeline on the main stream
  // (its tail is a latency-bound Horner chain that occupies a few dozen wavefronts)
  HIP_TRY(c, hipEventRecord(c->ev_fork, s));
  HIP_TRY(c, hipStreamWaitEvent(s2, c->ev_fork, 0));
  if (has_dyn) {
    Job dj = job;
    dj.d_st_scalars = nullptr; dj.d_st_index = nullptr; dj.d_st_offsets = nullptr; dj.n_static = 0;
    JobDesc jd;
    if (!c->forced_w && job.n_dyn <= 128ull * B) {
      // few proof-specific points per transaction: one wavefront per transaction, no global sort
      jd.w = 4; jd.n_windows = 64;
      c->last_w = 4;
      TRY(ensure(c, c->dyn_rows, std::max<uint64_t>(job.n_dyn, 1) * NIELS_WORDS * 4));
      TRY(ensure(c, c->window_sums, (size_t)B * 64 * EXT_WORDS * 4));
      TRY(ensure(c, c->window_flags, (size_t)B * 64 * 4));
      TRY(ensure(c, c->msm_fail, (size_t)B * 4));
      HIP_TRY(c, hipMemsetAsync(c->msm_fail.p, 0, (size_t)B * 4, s));
      {
        Launch l(c, "k_decompress");
        hipLaunchKernelGGL(k_decompress, dim3(blocks_for(job.n_dyn, 256)), dim3(256), 0, s, job.d_dyn_points,
                           (uint32_t*)c->dyn_rows.p, job.n_dyn, job.d_dyn_offsets, (uint32_t)B,
                           (uint32_t*)c->msm_fail.p, (unsigned long long*)((char*)c->status.p + 8), (uint8_t*)nullptr);
      }
      TRY(small_msm_launch(c, job, s));
    } else {
      TRY(run_to_windows(c, dj, jd, /*reset_status=*/false));
    }
    Launch l(c, "k_msm_finish_quad");
    hipLaunchKernelGGL(k_msm_finish_quad, dim3(blocks_for(4 * B, 256)), dim3(256), 0, s,
                       (const uint32_t*)c->window_sums.p, (const uint32_t*)c->window_flags.p,
                       (const uint32_t*)c->msm_fail.p, (uint8_t*)c->accept.p, (uint32_t*)nullptr,
                       (uint32_t*)c->dynsum.p, (uint32_t)B, jd.w, jd.n_windows, (const uint32_t*)nullptr, 0u);
  }
  if (job.n_static) {
    Launch l(c, "k_static_digits", s2);
    hipLaunchKernelGGL(k_static_digits, dim3(blocks_for(job.n_static, 256)), dim3(256), 0, s2, job.d_st_scalars,
                       (int16_t*)c->digits.p, job.n_static, ps->tbl_w, W, (uint32_t*)c->status.p,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, 0u);
  }
  {
    Launch l(c, "k_static_accumulate", s2);
    hipLaunchKernelGGL(k_static_accumulate<false>, dim3(blocks_for(n_lanes, 256)), dim3(256), 0, s2,
                       (const int16_t*)c->digits.p, job.d_st_offsets, job.d_st_index, (const uint32_t*)ps->table,
                       (uint32_t)ps->n, ps->tbl_H, W, P, (uint32_t)B, job.n_static, (uint32_t*)c->st_partials.p,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, 1u);
  }
  HIP_TRY(c, hipEventRecord(c->ev_join, s2));
  HIP_TRY(c, hipStreamWaitEvent(s, c->ev_join, 0));
  {
    Launch l(c, "k_static_combine");
    hipLaunchKernelGGL(k_static_combine, dim3(blocks_for((uint64_t)B * COMBINE_LANES, 64)), dim3(64), 0, s, (const uint32_t*)c->st_partials.p,
                       (uint32_t)(W * P), has_dyn ? (const uint32_t*)c->dynsum.p : (const uint32_t*)nullptr,
                       has_dyn ? (const uint8_t*)c->accept.p : (const uint8_t*)nullptr, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, (uint32_t)B, (uint8_t*)c->accept2.p, (uint32_t*)nullptr);
  }
  {
    Launch l(c, "k_pack_bitmap");
    hipLaunchKernelGGL(k_pack_bitmap, dim3(blocks_for(nbytes, 256)), dim3(256), 0, s, (const uint8_t*)c->accept2.p,
                       job.d_wellformed, (uint8_t*)c->bitmap.p, (uint32_t)B);
  }
  HIP_TRY(c, hipGetLastError());
  char* h = (char*)c->pinned;
  HIP_TRY(c, hipMemcpyAsync(h, c->bitmap.p, nbytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(h + nbytes, c->status.p, 16, hipMemcpyDeviceToHost, s));
  return ZKGPU_OK;
}

int batch_device_tables(zkgpu_ctx* c, const Job& job, const zkgpu_pointset* ps, uint8_t* accept_bitmap) {
  memset(accept_bitmap, 0, ((size_t)job.n_msm + 7) / 8);
  const int rc = batch_device_tables_enqueue(c, job, ps);
  if (rc != ZKGPU_OK) { c->split = zkgpu_ctx::SplitOp{}; return rc; }
  return batch_collect(c, accept_bitmap, nullptr);
}

// ---- pipelined submit / wait ---------------------------------------------------------------
// The kernels of one batch form a small DAG:
//     unpack -> transcript -> prepare -+-> decompress -> small_msm_windows -> finish -+-> combine -> pack
//                                      +-> static_digits -> static_accumulate --------+
// transcript and finish are latency-bound (a few dozen wavefronts walking long serial chains);
// prepare, small_msm_windows and static_accumulate fill the chip.  With several batches in flight
// the chip-filling kernels of ALL batches go through the two shared streams first-in first-out,
// while each batch's latency-bound kernels sit on its context's own light stream and run beside
// them.  (Letting whole batches share the chip on equal terms instead makes them finish together,
// start their transcripts together, and leave the chip idle for the length of a transcript.)
struct PrepLaunch {
  PrepShape sh;
  const uint32_t *d_init, *d_mono_chal, *d_mono_pow, *d_tgt_off, *d_term_q, *d_term_mono, *d_term_coef;
  const uint32_t* d_tape;
  uint32_t n_ops;
  const uint32_t *d_seg_info = nullptr, *d_seg_const = nullptr;   // cooperative transcript (n_seg = 0: not available)
  const uint16_t* d_seg_map = nullptr;
  uint32_t n_seg = 0;
  size_t lds_bytes;
  const uint32_t* d_com;
  const uint8_t* d_proofs;
  const uint32_t* d_r;
  size_t proof_len;
};

// proof bytes of a statement shape: the two-phase wire format, or the one-phase one (three points fewer)
inline bool proof_len_fits(const PrepShape& sh, size_t len) { return len == 1 + 4ull * sh.proof_words || len + 96 == 1 + 4ull * sh.proof_words; }
inline uint32_t proof_is_compact(const PrepShape& sh, size_t len) { return len + 96 == 1 + 4ull * sh.proof_words ? 1u : 0u; }

bool pipe_eligible(const zkgpu_ctx* c, const Job& job, const zkgpu_pointset* ps) {
  return ps && ps->table && job.n_static && job.n_dyn && !c->forced_w && job.n_dyn <= 128ull * job.n_msm;
}

int pipe_enqueue(zkgpu_ctx* c, const Job& job, const zkgpu_pointset* ps, const PrepLaunch* prep) {
  static_assert(std::is_trivially_copyable<Job>::value && std::is_trivially_copyable<PrepLaunch>::value, "kept as bytes");
  {
    std::vector<uint8_t> jb(sizeof(Job)), pb(prep ? sizeof(PrepLaunch) : 0);     // (job may alias c->last: copy first)
    memcpy(jb.data(), &job, sizeof(Job));
    if (prep) memcpy(pb.data(), prep, sizeof(PrepLaunch));
    c->last.job.swap(jb); c->last.prep.swap(pb);
    c->last.ps = ps; c->last.has_prep = prep != nullptr; c->last.valid = true;
  }
  const size_t B = job.n_msm;
  const size_t nbytes = (B + 7) / 8;
  hipStream_t L = c->stream_l, H1 = c->serial ? L : c->stream, H2 = c->serial ? L : c->stream2;
  const hipStream_t H3s = c->serial ? L : c->stream3;
  const int W = ps->tbl_W;
  int P = c->forced_parts;
  if (P <= 0) {
    P = (int)((131072 + B * W - 1) / (B * W));
    P = std::max(1, std::min(16, P));
  }
  const uint64_t n_lanes = (uint64_t)B * W * P;
  // group checks (k_group_combine): only for whole proofs (the weights come from k_transcript)
  const uint32_t group = (prep && c->group_size > 1) ? (uint32_t)std::min<size_t>(c->group_size, B) : 1;
  const uint32_t n_groups = (uint32_t)((B + group - 1) / group);
  const uint32_t ns = prep ? prep->sh.n_static : 0;
  // locating the culprit of a failed group saves work (one multiscalar multiplication instead of `group`) at the
  // price of two more dependent stages: worth it once the batch is large enough for the work to matter.  Mode 3 forms the
  // locating sums of ALL groups beside the group sums (twice the rows in the same multiplication), so that a failed group's
  // culprit is named by k_group_combine itself: no extra stage, ~3 % more point arithmetic per batch.
  const bool locate = group > 1 && (c->locate_mode >= 2 || (c->locate_mode == 0 && B >= LOCATE_MIN_BATCH));
  const bool spec = locate && c->locate_mode == 3;
  const uint32_t grp_rows = spec ? 2 * n_groups : n_groups;
  // parts per (check, window) of the group launch and of the individual re-check (few checks each:
  // short chains of additions per lane keep their latency down)
  int Pg = 1;
  const int Pf = 32;
  // the locating multiplication runs for the failed groups only, on the tail of the batch: many short chains (its grid is
  // sized for every group failing; lanes beyond the device-side count leave at once)
  const int Pl = c->locate_parts > 0 ? c->locate_parts : 32;
  if (group > 1) {
    Pg = (int)std::max<uint64_t>(1, std::min<uint64_t>(32, (65536 + (uint64_t)grp_rows * W - 1) / ((uint64_t)grp_rows * W)));
    TRY(ensure(c, c->grp_sc, (size_t)grp_rows * ns * 32));
    TRY(ensure(c, c->grp_digits, (size_t)grp_rows * ns * W * 2));
    TRY(ensure(c, c->grp_partials, (size_t)grp_rows * std::max<size_t>((size_t)W * std::max(Pg, locate && !spec ? Pl : 0), TAIL_THREADS) * EXT_WORDS * 4));
    TRY(ensure(c, c->grp_ok, n_groups));
    TRY(ensure(c, c->row_map, B * 4));
    TRY(ensure(c, c->grp_fail, (size_t)n_groups * 12));
    TRY(ensure(c, c->grp_fail_sum, (size_t)n_groups * EXT_WORDS * 4));
    TRY(ensure(c, c->grp_ws, (size_t)n_groups * 64 * EXT_WORDS * 4));
    TRY(ensure(c, c->grp_wf, (size_t)n_groups * 64 * 4));
    TRY(ensure(c, c->grp_dyn, (size_t)n_groups * EXT_WORDS * 4));
    TRY(ensure(c, c->rechk_pts, B * EXT_WORDS * 4));
  }
  TRY(ensure(c, c->accept, B));
  TRY(ensure(c, c->accept2, B));
  TRY(ensure(c, c->bitmap, nbytes));
  TRY(ensure_pinned(c, nbytes + 64));
  TRY(ensure(c, c->status, 64));
  TRY(ensure(c, c->digits, std::max<uint64_t>(job.n_static, 1) * W * 2));
  TRY(ensure(c, c->st_partials, std::max<uint64_t>(n_lanes, group > 1 ? (uint64_t)B * std::max(W * Pf, TAIL_THREADS) : 0) * EXT_WORDS * 4));
  TRY(ensure(c, c->dynsum, B * EXT_WORDS * 4));
  TRY(ensure(c, c->dyn_rows, std::max<uint64_t>(job.n_dyn, 1) * NIELS_WORDS * 4));
  TRY(ensure(c, c->window_sums, (size_t)B * 64 * EXT_WORDS * 4));
  TRY(ensure(c, c->window_flags, (size_t)B * 64 * 4));
  TRY(ensure(c, c->msm_fail, (size_t)B * 4));
  TRY(ensure(c, c->small_tbl, std::max<uint64_t>(job.n_dyn, 1) * SMALL_TBL * EXT_WORDS * 4));
  TRY(ensure(c, c->recoded, std::max<uint64_t>(job.n_dyn, 1) * 32));
  c->last_w = 4;
  if (c->reserve_only) return ZKGPU_OK;                 // (the workspace now fits a batch of this shape and size: nothing is launched)
  const int phase = prep ? c->enqueue_phase : (int)zkgpu_ctx::ENQ_ALL;        // (two halves: whole proofs only)
  if (phase == zkgpu_ctx::ENQ_ALL || phase == zkgpu_ctx::ENQ_FRONT) {
  {
    Launch l(c, "k_batch_init", L);
    hipLaunchKernelGGL(k_batch_init, dim3(blocks_for(B, 256)), dim3(256), 0, L, (uint32_t*)c->status.p,
                       (uint32_t*)c->msm_fail.p, prep ? (uint32_t*)c->prep_wf.p : (uint32_t*)nullptr, (uint32_t)B);
  }
  if (prep) {
    const PrepShape& sh = prep->sh;
    {
      Launch l(c, "k_proof_unpack", L);
      hipLaunchKernelGGL(k_proof_unpack, dim3(blocks_for((uint64_t)B * sh.proof_words, 256)), dim3(256), 0, L,
                         prep->d_proofs, (uint64_t)prep->proof_len, (uint32_t*)c->prep_pw.p, sh.proof_words,
                         (uint32_t)B, (uint32_t*)c->prep_wf.p, proof_is_compact(sh, prep->proof_len));
    }
    HIP_TRY(c, hipEventRecord(c->ev_u, L));
    // one wavefront per transaction while that still leaves the chip room (the cooperative form costs ~9x the
    // wave-instructions of the one-lane form, and buys latency only); beyond that, one lane per transaction
    const bool coop = prep->n_seg && (c->transcript_mode == 2 || (c->transcript_mode == 0 && B <= COOP_TRANSCRIPT_MAX));
    if (coop) {
      {
        Launch l(c, "k_tape_gather", L);
        hipLaunchKernelGGL(k_tape_gather, dim3(blocks_for((uint64_t)B * prep->n_seg * 25, 256)), dim3(256), 0, L, sh, prep->n_seg,
                           prep->d_seg_const, prep->d_seg_map, prep->d_com, (const uint32_t*)c->prep_pw.p, (uint32_t)B,
                           (uint2*)c->prep_absorb.p);
      }
      {
        Launch l(c, "k_transcript_coop", L);
        hipLaunchKernelGGL(k_transcript_coop, dim3((unsigned)B), dim3(64), 0, L, prep->n_seg, prep->d_seg_info, prep->d_init,
                           (const uint2*)c->prep_absorb.p, (uint32_t)B, sh.n_ch, (uint32_t*)c->prep_raw.p);
      }
      {
        Launch l(c, "k_challenges", L);
        hipLaunchKernelGGL(k_challenges, dim3((unsigned)B), dim3(128), (size_t)sh.n_ch * 32, L, sh, (const uint32_t*)c->prep_raw.p,
                           (const uint32_t*)c->prep_pw.p, prep->d_r, (uint32_t)B, (uint32_t*)c->prep_ch.p,
                           (uint32_t*)c->prep_wf.p, prep->d_mono_chal, prep->d_mono_pow, group > 1 ? 1u : 0u);
      }
    } else {
      Launch l(c, "k_transcript", L);
      hipLaunchKernelGGL(k_transcript, dim3(blocks_for(B, 64)), dim3(64), 0, L, sh, prep->d_init,
                         (const uint4*)prep->d_tape, prep->n_ops, prep->d_com, (const uint32_t*)c->prep_pw.p, prep->d_r, (uint32_t)B, (uint32_t*)c->prep_ch.p,
                         (uint32_t*)c->prep_wf.p, prep->d_mono_chal, prep->d_mono_pow, group > 1 ? 1u : 0u);
    }
  }
  HIP_TRY(c, hipEventRecord(c->ev_t, L));
  }                                                     // (front)
  if (phase == zkgpu_ctx::ENQ_FRONT) { HIP_TRY(c, hipGetLastError()); c->awaiting_back = true; return ZKGPU_OK; }
  if (prep && (phase == zkgpu_ctx::ENQ_ALL || phase == zkgpu_ctx::ENQ_MID)) {
    // the proof-specific points need the proof bytes only: gather, decompress and build their small tables on the shared
    // stream while the transcript is replayed.  Queued AFTER the transcript: k_points_tables takes every register of the chip
    // (255 per lane, two wavefronts per SIMD), and a light kernel queued behind it waits for one of its wavefronts to retire
    // -- ~0.6 ms (measured: the next batch's 30 us merge copy took 0.59 ms there, profiles/archive/r04an_timeline.txt)
    const PrepShape& sh = prep->sh;
    HIP_TRY(c, hipStreamWaitEvent(H1, c->ev_u, 0));
    {
      Launch l(c, "k_points_tables", H1);
      hipLaunchKernelGGL(k_points_tables, dim3(blocks_for((uint64_t)B * sh.n_dyn, 256)), dim3(256), 0, H1, sh,
                         prep->d_com, (const uint32_t*)c->prep_pw.p, (uint32_t)B, (uint32_t*)c->small_tbl.p,
                         (uint32_t*)c->msm_fail.p, (unsigned long long*)((char*)c->status.p + 8));
    }
    HIP_TRY(c, hipEventRecord(c->ev_dig, H1));      // msm_fail is final: the group sums leave such transactions out
  }
  if (phase == zkgpu_ctx::ENQ_MID) { HIP_TRY(c, hipGetLastError()); return ZKGPU_OK; }
  c->awaiting_back = false;
  if (prep) {
    hipStream_t H3 = H3s;
    HIP_TRY(c, hipStreamWaitEvent(H3, c->ev_t, 0));
    Launch l(c, "k_prepare", H3);
    hipLaunchKernelGGL(k_prepare, dim3((unsigned)B), dim3(256), prep->lds_bytes, H3, prep->sh, prep->d_mono_chal,
                       prep->d_mono_pow, prep->d_tgt_off, prep->d_term_q, (const uint2*)prep->d_term_mono, prep->d_term_coef,
                       (const uint32_t*)c->prep_ch.p, prep->d_com, (const uint32_t*)c->prep_pw.p,
                       (uint32_t*)c->prep_dyn_sc.p, (uint32_t*)c->recoded.p, (uint32_t*)c->prep_st_sc.p);
  }
  HIP_TRY(c, hipEventRecord(c->ev_p, prep ? H3s : L));
  HIP_TRY(c, hipStreamWaitEvent(H1, c->ev_p, 0));
  HIP_TRY(c, hipStreamWaitEvent(H2, c->ev_p, 0));
  uint32_t* n_recheck = (uint32_t*)((char*)c->status.p + 32);
  if (group > 1) {
    HIP_TRY(c, hipStreamWaitEvent(H2, c->ev_dig, 0));
    {
      Launch l(c, "k_group_scalars", H2);     // sums and their digits in one launch
      hipLaunchKernelGGL(k_group_scalars, dim3(blocks_for((uint64_t)n_groups * ns, 256)), dim3(256), 0, H2,
                         job.d_st_scalars, (uint32_t)B, ns, group, (uint32_t*)c->grp_sc.p, (int16_t*)c->grp_digits.p,
                         ps->tbl_w, W, (const uint32_t*)c->msm_fail.p, job.d_wellformed, spec ? 1u : 0u);
    }
    {
      Launch l(c, "k_static_accumulate", H2);
      hipLaunchKernelGGL(k_static_accumulate<false>, dim3(blocks_for((uint64_t)grp_rows * W * Pg, 256)), dim3(256), 0, H2,
                         (const int16_t*)c->grp_digits.p, job.d_st_offsets, job.d_st_index, (const uint32_t*)ps->table,
                         (uint32_t)ps->n, ps->tbl_H, W, Pg, grp_rows, (uint64_t)grp_rows * ns,
                         (uint32_t*)c->grp_partials.p, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 1u);
    }
    HIP_TRY(c, hipEventRecord(c->ev_sa, H2));
  } else {
    {
      Launch l(c, "k_static_digits", H2);
      hipLaunchKernelGGL(k_static_digits, dim3(blocks_for(job.n_static, 256)), dim3(256), 0, H2, job.d_st_scalars,
                         (int16_t*)c->digits.p, job.n_static, ps->tbl_w, W, (uint32_t*)c->status.p,
                         (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, 0u);
    }
    {
      Launch l(c, "k_static_accumulate", H2);
      hipLaunchKernelGGL(k_static_accumulate<false>, dim3(blocks_for(n_lanes, 256)), dim3(256), 0, H2,
                         (const int16_t*)c->digits.p, job.d_st_offsets, job.d_st_index, (const uint32_t*)ps->table,
                         (uint32_t)ps->n, ps->tbl_H, W, P, (uint32_t)B, job.n_static, (uint32_t*)c->st_partials.p,
                         (const uint32_t*)nullptr, (const uint32_t*)nullptr, 1u);
    }
    HIP_TRY(c, hipEventRecord(c->ev_sa, H2));
  }
  if (prep) {
    TRY(small_accumulate_launch(c, job, H1));       // points and tables were done beside the transcript
  } else {
    {
      Launch l(c, "k_decompress", H1);
      hipLaunchKernelGGL(k_decompress, dim3(blocks_for(job.n_dyn, 256)), dim3(256), 0, H1, job.d_dyn_points,
                         (uint32_t*)c->dyn_rows.p, job.n_dyn, job.d_dyn_offsets, (uint32_t)B, (uint32_t*)c->msm_fail.p,
                         (unsigned long long*)((char*)c->status.p + 8), (uint8_t*)nullptr);
    }
    TRY(small_msm_launch(c, job, H1));
  }
  HIP_TRY(c, hipEventRecord(c->ev_sm, H1));
  HIP_TRY(c, hipStreamWaitEvent(L, c->ev_sm, 0));
  // Horner chains (the longest dependent chain of a batch): one per GROUP over the summed windows of its transactions,
  // the transactions of a failed group getting theirs after k_group_combine has named the group -- a third less point
  // arithmetic, but a batch in which ANY group fails walks two chains in a row, and the chain is latency, not work
  // (measured: +2 % without failures, -3 % with 1 bad transaction in 64).  So: per group while the batch this context
  // family finished last had no failed group at all (root->last_failed_groups), else one chain per transaction up
  // front.  Same verdicts either way.
  zkgpu_ctx* root = c->parent ? c->parent : c;
  const bool group_first = group > 1 && !spec && (c->horner_mode == 2 || (c->horner_mode == 0 && root->last_failed_groups.load() == 0));
  if (group_first) {
    {
      Launch l(c, "k_group_windows", L);
      hipLaunchKernelGGL(k_group_windows, dim3(blocks_for((uint64_t)n_groups * 64, 256)), dim3(256), 0, L,
                         (const uint32_t*)c->window_sums.p, (const uint32_t*)c->window_flags.p, (const uint32_t*)c->msm_fail.p,
                         job.d_wellformed, (uint32_t)B, group, 64u, (uint32_t*)c->grp_ws.p, (uint32_t*)c->grp_wf.p);
    }
    Launch l(c, "k_msm_finish_quad", L);
    hipLaunchKernelGGL(k_msm_finish_quad, dim3(blocks_for(4 * (uint64_t)n_groups, 256)), dim3(256), 0, L,
                       (const uint32_t*)c->grp_ws.p, (const uint32_t*)c->grp_wf.p, (const uint32_t*)nullptr, (uint8_t*)nullptr,
                       (uint32_t*)nullptr, (uint32_t*)c->grp_dyn.p, n_groups, 4, 64, (const uint32_t*)nullptr, 0u);
  } else {
    Launch l(c, "k_msm_finish_quad", L);
    hipLaunchKernelGGL(k_msm_finish_quad, dim3(blocks_for(4 * B, 256)), dim3(256), 0, L,
                       (const uint32_t*)c->window_sums.p, (const uint32_t*)c->window_flags.p,
                       (const uint32_t*)c->msm_fail.p, (uint8_t*)c->accept.p, (uint32_t*)nullptr,
                       (uint32_t*)c->dynsum.p, (uint32_t)B, 4, 64, (const uint32_t*)nullptr, 0u);
  }
  HIP_TRY(c, hipStreamWaitEvent(L, c->ev_sa, 0));
  if (group > 1) {
    uint32_t* n_fail = (uint32_t*)((char*)c->status.p + 36);
    uint32_t* fail_list = (uint32_t*)c->grp_fail.p;
    uint32_t* cand = fail_list + n_groups;
    uint32_t* grp_state = cand + n_groups;
    {
      Launch l(c, "k_group_combine", L);        // verdicts of the groups; a failed group also gets its locating scalars
      hipLaunchKernelGGL(spec ? k_group_combine<true> : k_group_combine<false>, dim3(n_groups), dim3(256), 0, L, (const uint32_t*)c->grp_partials.p,
                         (uint32_t)(W * Pg), group_first ? (const uint32_t*)c->grp_dyn.p : (const uint32_t*)nullptr,
                         (const uint32_t*)c->dynsum.p, (const uint32_t*)c->msm_fail.p,
                         job.d_wellformed, (uint32_t)B, group, (uint8_t*)c->accept2.p, grp_state, fail_list,
                         (uint32_t*)c->grp_fail_sum.p, n_fail, job.d_st_scalars, ns, (uint32_t*)c->grp_sc.p,
                         (int16_t*)c->grp_digits.p, ps->tbl_w, W, spec ? 2u : locate ? 1u : 0u, (uint32_t*)c->row_map.p, n_recheck, cand,
                         (int16_t*)c->digits.p);
    }
    if (group_first) {
      Launch l(c, "k_msm_finish_quad", L);      // Horner chains of the transactions of the failed groups only
      hipLaunchKernelGGL(k_msm_finish_quad, dim3(blocks_for(4 * B, 256)), dim3(256), 0, L,
                         (const uint32_t*)c->window_sums.p, (const uint32_t*)c->window_flags.p,
                         (const uint32_t*)c->msm_fail.p, (uint8_t*)c->accept.p, (uint32_t*)nullptr,
                         (uint32_t*)c->dynsum.p, (uint32_t)B, 4, 64, (const uint32_t*)grp_state, group);
    }
    // failed groups (kernels.hpp, "group checks"): one more multiscalar multiplication each LOCATES the bad
    // transaction, which alone is then checked on its own.  Grids are sized for the worst case; lanes beyond the
    // device-side counts leave at once (no failed group: four near-empty launches).
    const bool fused_tail = c->tail_mode == 0;
    if (locate && !spec && fused_tail) {
      Launch l(c, "k_locate_fused", L);
      hipLaunchKernelGGL(k_locate_fused, dim3(n_groups), dim3(LOCATE_THREADS), 0, L, (const int16_t*)c->grp_digits.p, job.d_st_offsets, job.d_st_index,
                         (const uint32_t*)ps->table, (uint32_t)ps->n, ps->tbl_H, n_groups, (uint32_t*)c->grp_partials.p,
                         (const uint32_t*)c->dynsum.p, (const uint32_t*)c->msm_fail.p, job.d_wellformed, (uint32_t)B, group,
                         (const uint32_t*)fail_list, (const uint32_t*)n_fail, (const uint32_t*)c->grp_fail_sum.p,
                         (uint32_t*)c->row_map.p, n_recheck, cand, job.d_st_scalars, ns, (int16_t*)c->digits.p, ps->tbl_w, W);
    }
    if (locate && !spec && !fused_tail) {
      Launch l(c, "k_static_accumulate", L);
      hipLaunchKernelGGL(k_static_accumulate<false>, dim3(blocks_for((uint64_t)n_groups * W * Pl, 256)), dim3(256), 0, L,
                         (const int16_t*)c->grp_digits.p, job.d_st_offsets, job.d_st_index, (const uint32_t*)ps->table,
                         (uint32_t)ps->n, ps->tbl_H, W, Pl, n_groups, (uint64_t)n_groups * ns,
                         (uint32_t*)c->grp_partials.p, (const uint32_t*)nullptr, (const uint32_t*)n_fail, 1u);
    }
    if (!locate) {
      Launch l(c, "k_static_digits", L);        // digits of the queued transactions only
      hipLaunchKernelGGL(k_static_digits, dim3(blocks_for(job.n_static, 256)), dim3(256), 0, L, job.d_st_scalars,
                         (int16_t*)c->digits.p, job.n_static, ps->tbl_w, W, (uint32_t*)c->status.p,
                         (const uint32_t*)c->row_map.p, (const uint32_t*)n_recheck, ns, 0u);
    } else if (!spec && !fused_tail) {
      Launch l(c, "k_locate_combine", L);       // names the culprit (or queues the whole group) and writes the digits of the queued
      hipLaunchKernelGGL(k_locate_combine, dim3(n_groups), dim3(256), 0, L, (const uint32_t*)c->grp_partials.p,
                         (uint32_t)(W * Pl), (const uint32_t*)c->dynsum.p, (const uint32_t*)c->msm_fail.p, job.d_wellformed,
                         (uint32_t)B, group, (const uint32_t*)fail_list, (const uint32_t*)n_fail,
                         (const uint32_t*)c->grp_fail_sum.p, (uint32_t*)c->row_map.p, n_recheck, cand, job.d_st_scalars, ns,
                         (int16_t*)c->digits.p, ps->tbl_w, W);
    }
    if (fused_tail) {
      Launch l(c, "k_recheck_fused", L);
      hipLaunchKernelGGL(k_recheck_fused, dim3((unsigned)B), dim3(TAIL_THREADS), 0, L, (const int16_t*)c->digits.p, job.d_st_offsets, job.d_st_index,
                         (const uint32_t*)ps->table, (uint32_t)ps->n, ps->tbl_H, W, (uint64_t)job.n_static, (uint32_t*)c->st_partials.p,
                         (const uint32_t*)c->dynsum.p, (const uint8_t*)c->accept.p, (const uint32_t*)c->row_map.p,
                         (const uint32_t*)n_recheck, (uint8_t*)c->accept2.p, (uint32_t*)c->rechk_pts.p);
    } else {
      {
      Launch l(c, "k_static_accumulate", L);
      hipLaunchKernelGGL(k_static_accumulate<false>, dim3(blocks_for((uint64_t)B * W * Pf, 256)), dim3(256), 0, L,
                         (const int16_t*)c->digits.p, job.d_st_offsets, job.d_st_index, (const uint32_t*)ps->table,
                         (uint32_t)ps->n, ps->tbl_H, W, Pf, (uint32_t)B, job.n_static, (uint32_t*)c->st_partials.p,
                         (const uint32_t*)c->row_map.p, (const uint32_t*)n_recheck, 1u);
    }
    {
      Launch l(c, "k_static_combine", L);
      hipLaunchKernelGGL(k_static_combine, dim3(blocks_for((uint64_t)B * COMBINE_LANES, 64)), dim3(64), 0, L, (const uint32_t*)c->st_partials.p,
                         (uint32_t)(W * Pf), (const uint32_t*)c->dynsum.p, (const uint8_t*)c->accept.p,
                         (const uint32_t*)c->row_map.p, (const uint32_t*)n_recheck, (uint32_t)B, (uint8_t*)c->accept2.p,
                         (uint32_t*)c->rechk_pts.p);
    }
    }
    {
      Launch l(c, "k_pack_bitmap_groups", L);   // with the verdict of the located groups' other transactions: S1 - E_b
      hipLaunchKernelGGL(k_pack_bitmap_groups, dim3(blocks_for(nbytes, 256)), dim3(256), 0, L, (const uint8_t*)c->accept2.p,
                         job.d_wellformed, (const uint32_t*)c->msm_fail.p, (uint8_t*)c->bitmap.p, (uint32_t)B, group,
                         (const uint32_t*)grp_state, (const uint32_t*)c->grp_fail_sum.p, (const uint32_t*)cand,
                         (const uint32_t*)c->row_map.p, (const uint32_t*)c->rechk_pts.p, (uint32_t*)c->status.p,
                         c->force_unresolved ? 1u : 0u);
    }
  } else {
    Launch l(c, "k_static_combine", L);
    hipLaunchKernelGGL(k_static_combine, dim3(blocks_for((uint64_t)B * COMBINE_LANES, 64)), dim3(64), 0, L, (const uint32_t*)c->st_partials.p,
                       (uint32_t)(W * P), (const uint32_t*)c->dynsum.p, (const uint8_t*)c->accept.p,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t)B, (uint8_t*)c->accept2.p, (uint32_t*)nullptr);
  }
  if (group <= 1) {
    Launch l(c, "k_pack_bitmap", L);
    hipLaunchKernelGGL(k_pack_bitmap, dim3(blocks_for(nbytes, 256)), dim3(256), 0, L, (const uint8_t*)c->accept2.p,
                       job.d_wellformed, (uint8_t*)c->bitmap.p, (uint32_t)B);
  }
  HIP_TRY(c, hipGetLastError());
  char* h = (char*)c->pinned;
  HIP_TRY(c, hipMemcpyAsync(h, c->bitmap.p, nbytes, hipMemcpyDeviceToHost, L));
  HIP_TRY(c, hipMemcpyAsync(h + nbytes, c->status.p, 48, hipMemcpyDeviceToHost, L));
  HIP_TRY(c, hipEventRecord(c->ev_done, L));
  c->pending = true;
  c->pending_batch = B;
  c->sync_result_valid = false;
  return ZKGPU_OK;
}

// a submit whose shape the pipeline does not cover has already run synchronously: park its result
void park_sync_result(zkgpu_ctx* c, int rc, const uint8_t* bitmap, size_t batch) {
  c->sync_result.assign(bitmap, bitmap + (batch + 7) / 8);
  c->sync_rc = rc;
  c->sync_result_valid = true;
  c->pending = true;
  c->pending_batch = batch;
}

int pipe_wait(zkgpu_ctx* c, uint8_t* accept_bitmap) {
  if (!c->pending) return ZKGPU_EINVAL;
  const size_t nbytes = (c->pending_batch + 7) / 8;
  c->pending = false;
  memset(accept_bitmap, 0, nbytes);
  if (c->sync_result_valid) {
    c->sync_result_valid = false;
    if (c->sync_rc == ZKGPU_OK) memcpy(accept_bitmap, c->sync_result.data(), nbytes);
    return c->sync_rc;
  }
  HIP_TRY(c, hipEventSynchronize(c->ev_done));
  if (c->profiling) prof_collect(c);
  const char* h = (const char*)c->pinned;
  uint32_t st;
  memcpy(&st, h + nbytes, 4);
  if (c->group_size > 1) {             // failed groups steer the next batches' Horner arrangement (pipe_enqueue)
    uint32_t n_fail;
    memcpy(&n_fail, h + nbytes + 36, 4);
    (c->parent ? c->parent : c)->last_failed_groups.store(n_fail);
  }
  if ((st & 4u) && c->last.valid && c->group_size > 1) {
    // a located transaction did not account for its group's sum (probability ~2^-248, or the test hook): verdicts
    // must not rest on it -- run the same batch again with every transaction checked on its own
    Job job;
    PrepLaunch prep;
    memcpy(&job, c->last.job.data(), sizeof job);
    if (c->last.has_prep) memcpy(&prep, c->last.prep.data(), sizeof prep);
    const int saved = c->group_size;
    c->group_size = 1;
    const int rc = pipe_enqueue(c, job, c->last.ps, c->last.has_prep ? &prep : nullptr);
    c->group_size = saved;
    c->pending = false;
    ++c->regroup_fallbacks;
    if (rc != ZKGPU_OK) return rc;
    HIP_TRY(c, hipEventSynchronize(c->ev_done));
    if (c->profiling) prof_collect(c);
    memcpy(&st, h + nbytes, 4);
  }
  if (st & 2u) { c->last_error = "scalar with bit 255 set"; return ZKGPU_EINVAL; }
  memcpy(accept_bitmap, h, nbytes);
  return ZKGPU_OK;
}

int upload(zkgpu_ctx* c, Buffer& b, const void* src, size_t bytes, hipStream_t stream = nullptr) {
  TRY(ensure(c, b, std::max<size_t>(bytes, 16)));
  if (bytes) HIP_TRY(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, stream ? stream : c->stream));
  return ZKGPU_OK;
}

bool offsets_ok(const uint64_t* off, size_t batch, uint64_t* total) {
  if (off[0] != 0) return false;
  for (size_t i = 0; i < batch; ++i) if (off[i + 1] < off[i]) return false;
  *total = off[batch];
  return true;
}

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { (void)hipGetDevice(&prev); if (prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

// =============================== C ABI =========================================
extern "C" {

int zkgpu_abi_version(void) { return 3; }

const char* zkgpu_strerror(int code) {
  switch (code) {
    case ZKGPU_OK: return "ok";
    case ZKGPU_EINVAL: return "invalid argument";
    case ZKGPU_EINVALID_POINT: return "invalid ristretto255 encoding";
    case ZKGPU_EHIP: return "HIP runtime error";
    case ZKGPU_ENOMEM: return "out of device memory";
    case ZKGPU_ENODEVICE: return "no usable HIP device";
    case ZKGPU_ENOCOMM: return "RCCL unavailable or a collective failed";
    case ZKGPU_EREMOTE: return "another rank of the sharded verification failed";
    case ZKGPU_WSECOND_VERIFIER: return "created, but another verifier is alive on this device with 20 or more hardware queues (warning)";
    default: return "unknown error";
  }
}

const char* zkgpu_last_error(const zkgpu_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

namespace {
bool g_hw_queues_late = false;     // GPU_MAX_HW_QUEUES was unset when the HIP runtime started (seen by zkgpu_runtime_hint or the first zkgpu_init)

// is the compute driver's device node open in this process (= has the HIP / HSA runtime started)?
bool kfd_is_open() {
  DIR* d = opendir("/proc/self/fd");
  if (!d) return false;
  bool found = false;
  while (struct dirent* e = readdir(d)) {
    char path[64], target[64];
    snprintf(path, sizeof path, "/proc/self/fd/%s", e->d_name);
    const ssize_t n = readlink(path, target, sizeof target - 1);
    if (n > 0) { target[n] = 0; if (strcmp(target, "/dev/kfd") == 0) { found = true; break; } }
  }
  closedir(d);
  return found;
}

// Do kernels queued on `a` and on `b` run at the same time?  Two ~80 us spinning wavefronts, one per stream, released
// together: side by side both finish after ~one spin, on a shared hardware queue the second finishes after two.
// -1: the probe itself failed.  (Idle device assumed: called at creation time.)
int streams_overlap(hipStream_t a, hipStream_t b) {
  if (a == b) return 0;
  hipEvent_t e0 = nullptr, ea = nullptr, eb = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) return -1;
  const unsigned long long cycles = 200000;
  int verdict = -1;
  for (int attempt = 0; attempt < 2; ++attempt) {               // the first round also pays both streams' first launch
    hipError_t e = hipEventRecord(e0, a);
    if (e == hipSuccess) e = hipStreamWaitEvent(b, e0, 0);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, cycles, (uint32_t*)nullptr);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, cycles, (uint32_t*)nullptr);
    if (e == hipSuccess) e = hipEventRecord(ea, a);
    if (e == hipSuccess) e = hipEventRecord(eb, b);
    if (e == hipSuccess) e = hipEventSynchronize(ea);
    if (e == hipSuccess) e = hipEventSynchronize(eb);
    float ta = 0, tb = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ta, e0, ea);
    if (e == hipSuccess) e = hipEventElapsedTime(&tb, e0, eb);
    if (e != hipSuccess) { verdict = -1; break; }
    const float lo = ta < tb ? ta : tb, hi = ta < tb ? tb : ta;
    verdict = hi < 1.6f * lo ? 1 : 0;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
  return verdict;
}

// aux: a context for the general (one-stream-pair) paths only -- two streams of its own, no light stream, no third one:
// what the key and signature stages of zkgpu_tx_verify_batch run on beside the verifier's lanes (session.hpp)
// aux: 0 a root or forked context, 1 an auxiliary one (two high-priority streams, no light stream), 2 a prover slice (one stream)
int ctx_create(int device, zkgpu_ctx* parent, zkgpu_ctx** out, int aux = 0) {
  zkgpu_ctx* c = new zkgpu_ctx();
  c->device = device;
  // main stream: the proof-point pipeline, high priority; stream2: the chip-filling generator
  // kernel, low priority; stream_l: latency-bound kernels (transcripts, Horner tails), high priority
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  bool ok = true;
  if (parent) {
    constexpr int lanes = STREAM_SETS;
    ++parent->n_forks;
    const int set = (int)(++parent->fork_seq % lanes);
    c->parent = parent;
    while (ok && (int)parent->lane_streams.size() < 3 * (lanes - 1)) {
      hipStream_t st = nullptr;
      const int which = (int)parent->lane_streams.size() % 3;
      ok = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, which == 1 ? prio_least : prio_greatest) == hipSuccess;
      if (ok) parent->lane_streams.push_back(st);
    }
    if (ok && set > 0) {
      c->stream = parent->lane_streams[3 * (set - 1)];
      c->stream2 = parent->lane_streams[3 * (set - 1) + 1];
      c->stream3 = parent->lane_streams[3 * (set - 1) + 2];
    } else {
      c->stream = parent->stream;
      c->stream2 = parent->stream2;
      c->stream3 = parent->stream3;
    }
    c->owns_streams = false;
    c->group_size = parent->group_size;
    c->transcript_mode = parent->transcript_mode;
    c->locate_mode = parent->locate_mode;
    c->locate_parts = parent->locate_parts;
    c->tail_mode = parent->tail_mode;
    c->horner_mode = parent->horner_mode;
  } else if (aux == 2) {
    ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest) == hipSuccess;
  } else if (aux) {
    ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest) == hipSuccess &&
         hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_greatest) == hipSuccess;
  } else {
    ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest) == hipSuccess &&
         hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_least) == hipSuccess &&
         hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_greatest) == hipSuccess;
    // stream and stream2 carry the two halves of every pipeline that forks (decompression beside the digit sort of a large
    // multiscalar multiplication; the generator terms beside the proof points): if the runtime put them on one hardware
    // queue they would take turns.  Probe, and ask for another stream2 a few times if so (each new stream lands on the next
    // queue of the runtime's round robin).
    std::vector<hipStream_t> rejected;
    for (int attempt = 0; ok && attempt < 6; ++attempt) {
      const int ov = streams_overlap(c->stream, c->stream2);
      c->pair_overlaps = ov;
      if (ov != 0) break;
      rejected.push_back(c->stream2);
      c->stream2 = nullptr;
      ok = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_least) == hipSuccess;
    }
    for (hipStream_t st : rejected) (void)hipStreamDestroy(st);
  }
  if (!aux) ok = ok && hipStreamCreateWithPriority(&c->stream_l, hipStreamNonBlocking, prio_greatest) == hipSuccess;
  hipEvent_t* evs[] = {&c->ev_fork, &c->ev_join, &c->ev_t, &c->ev_p, &c->ev_sm, &c->ev_sa, &c->ev_done, &c->ev_dig, &c->ev_u};
  for (hipEvent_t* e : evs) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  if (!ok) { if (parent) --parent->n_forks; delete c; return ZKGPU_EHIP; }
  *out = c;
  return ZKGPU_OK;
}
}  // namespace

// What the host should have exported before its first HIP call.  18, not more: the runtime keeps that many queues PER
// PRIORITY; a verifier's high-priority streams take all of them, its low-priority and default-priority streams (2 + 2) come on
// top, and from 25 queues in all the device no longer runs them side by side -- which ones wait is a per-process lottery
// (measured, profiles/archive/r04v_tx_hwq.txt, r04w_hwq_bench.txt: a 32 768-transaction call at 16.1-18.8 ms in every process
// with 18; 16.5-18.5 OR 23-53 ms, one process in four, with 20 / 24; headline, steady state and config 4 equal at 16 / 18 / 24).
// The variable is read ONCE, when the HIP runtime starts (the kernel driver's device node opened in this process): after
// that, setting it changes nothing -- answer 2, remembered, and reported by zkgpu_ctx_queue_info and zkgpu_verifier_create.
int zkgpu_runtime_hint(char* buf, size_t cap) {
  static const char kHint[] = "GPU_MAX_HW_QUEUES=18";
  if (buf && cap) { strncpy(buf, kHint, cap - 1); buf[cap - 1] = 0; }
  if (getenv("GPU_MAX_HW_QUEUES")) return g_hw_queues_late ? ZKGPU_HINT_LATE : ZKGPU_HINT_PRESENT;
  if (kfd_is_open()) { g_hw_queues_late = true; return ZKGPU_HINT_LATE; }
  return ZKGPU_HINT_APPLY;
}

int zkgpu_init(int device, zkgpu_ctx** out) {
  if (!out) return ZKGPU_EINVAL;
  *out = nullptr;
  // Batches in flight use a stream each plus six shared ones; the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
  // queues (4 unless the process exported more BEFORE its first HIP call), and streams that share a queue while waiting on
  // each other's events crawl.  The library does NOT edit the process environment: the host asks
  // zkgpu_runtime_hint() what to export and exports it itself.  Here it is only noticed when nobody did: the process then runs
  // on the runtime's default, reported by zkgpu_ctx_queue_info / zkgpu_verifier_create.  DESIGN.md sec 5.1.
  {
    static std::once_flag once;
    std::call_once(once, [] { if (!getenv("GPU_MAX_HW_QUEUES")) g_hw_queues_late = true; });
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ZKGPU_ENODEVICE;
  if (hipSetDevice(device) != hipSuccess) return ZKGPU_ENODEVICE;
  return ctx_create(device, nullptr, out);
}

// Plain device memory for callers that have no HIP binding of their own (the *_dev entry points
// take what these return).
int zkgpu_malloc(zkgpu_ctx* c, size_t bytes, void** out) {
  if (!c || !out) return ZKGPU_EINVAL;
  *out = nullptr;
  DeviceGuard g(c->device);
  hipError_t e = hipMalloc(out, std::max<size_t>(bytes, 16));
  if (e != hipSuccess) { c->last_error = std::string("hipMalloc: ") + hipGetErrorString(e); *out = nullptr; return ZKGPU_ENOMEM; }
  return ZKGPU_OK;
}

int zkgpu_free(zkgpu_ctx* c, void* p) {
  if (!c) return ZKGPU_EINVAL;
  if (!p) return ZKGPU_OK;
  DeviceGuard g(c->device);
  HIP_TRY(c, hipFree(p));
  return ZKGPU_OK;
}

int zkgpu_upload(zkgpu_ctx* c, void* d_dst, const void* src, size_t bytes) {
  if (!c || (bytes && (!d_dst || !src))) return ZKGPU_EINVAL;
  DeviceGuard g(c->device);
  if (bytes) HIP_TRY(c, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
  return ZKGPU_OK;
}

int zkgpu_ctx_fork(zkgpu_ctx* parent, zkgpu_ctx** out) {
  if (!parent || !out) return ZKGPU_EINVAL;
  *out = nullptr;
  // every fork holds a hardware queue for its light stream, every set of shared streams three more;
  // past ~22 queues per process the runtime multiplexes them in software and streams that wait on
  // each other's events crawl (measured: 200 ms per step instead of 0.7)
  std::lock_guard<std::recursive_mutex> lk(parent->mu);      // n_forks and the lane streams belong to the parent
  if (parent->n_forks >= MAX_FORKS) {
    parent->last_error = "too many forks of one context (each batch in flight holds a hardware queue)";
    return ZKGPU_EINVAL;
  }
  DeviceGuard g(parent->device);
  return ctx_create(parent->device, parent, out);
}

void zkgpu_destroy(zkgpu_ctx* c) {
  if (!c) return;
  for (zkgpu_ctx* t : c->pv_slices) zkgpu_destroy(t);
  c->pv_slices.clear();
  DeviceGuard g(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  if (c->stream_l) (void)hipStreamSynchronize(c->stream_l);
  Buffer* bufs[] = {&c->in_scalars, &c->in_points, &c->in_offsets, &c->in_st_scalars, &c->in_st_index,
                    &c->in_st_offsets, &c->dyn_rows, &c->bins, &c->block_sums, &c->entries, &c->buckets,
                    &c->partials, &c->partial_flags, &c->window_sums, &c->window_flags, &c->msm_fail,
                    &c->status, &c->accept, &c->bitmap, &c->ok_bytes, &c->values, &c->uniform,
                    &c->digits, &c->st_partials, &c->dynsum, &c->accept2, &c->bin_order, &c->class_count, &c->part_hist, &c->part_entries, &c->part_lo, &c->dec_scratch, &c->heavy, &c->small_tbl, &c->recoded, &c->grp_sc, &c->grp_digits, &c->grp_partials, &c->grp_ok, &c->row_map, &c->grp_fail, &c->grp_fail_sum, &c->rechk_pts, &c->grp_ws, &c->grp_wf, &c->grp_dyn, &c->pv_plan, &c->pv_state, &c->pv_in, &c->pv_rows0, &c->pv_rows1, &c->pv_rows2, &c->pv_rows3, &c->pv_lay, &c->pv_pts, &c->pv_com, &c->pv_ab, &c->pv_proofs, &c->prep_com, &c->prep_proofs, &c->prep_r,
                    &c->prep_pw, &c->prep_ch, &c->prep_wf, &c->prep_dyn_sc, &c->prep_dyn_pt, &c->prep_st_sc,
                    &c->prep_absorb, &c->prep_raw, &c->ipa_lv, &c->ipa_rv, &c->ipa_cg, &c->ipa_ch, &c->ipa_w, &c->ipa_u,
                    &c->coal_com, &c->coal_proofs, &c->coal_r};
  for (Buffer* b : bufs) if (b->p) (void)hipFree(b->p);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->pinned_in) (void)hipHostFree(c->pinned_in);
  for (auto& e : c->ev_pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  if (c->owns_streams) {
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    for (hipStream_t st : c->lane_streams) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  }
  if (c->stream_l) (void)hipStreamDestroy(c->stream_l);
  hipEvent_t evs[] = {c->ev_fork, c->ev_join, c->ev_t, c->ev_p, c->ev_sm, c->ev_sa, c->ev_done, c->ev_dig, c->ev_u};
  for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
  if (c->parent) { std::lock_guard<std::recursive_mutex> lk(c->parent->mu); --c->parent->n_forks; }
  delete c;
}

int zkgpu_msm_dev(zkgpu_ctx* c, const void* d_scalars, const void* d_points, size_t n, uint8_t out[32],
                  size_t* bad_index) {
  if (!c || !out || (n && (!d_scalars || !d_points))) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  memset(out, 0, 32);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  int rc = msm_device(c, d_scalars, d_points, n, out, bad_index);
  if (rc != ZKGPU_OK) memset(out, 0, 32);
  return rc;
}

int zkgpu_msm(zkgpu_ctx* c, const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t out[32],
              size_t* bad_index) {
  if (!c || !out || (n && (!scalars || !points))) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  memset(out, 0, 32);
  TRY(refuse_if_pending(c));
  TRY(upload(c, c->in_scalars, scalars, n * 32));
  TRY(upload(c, c->in_points, points, n * 32));
  int rc = msm_device(c, c->in_scalars.p, c->in_points.p, n, out, bad_index);
  if (rc != ZKGPU_OK) memset(out, 0, 32);
  return rc;
}

int zkgpu_verify_batch_dev(zkgpu_ctx* c, const void* d_scalars, const void* d_points, const void* d_offsets,
                           size_t batch, size_t n_terms, uint8_t* accept_bitmap) {
  if (!c || !accept_bitmap || (batch && !d_offsets) || (n_terms && (!d_scalars || !d_points))) return ZKGPU_EINVAL;
  if (batch >= (1ull << 31)) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  memset(accept_bitmap, 0, (batch + 7) / 8);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  Job job;
  job.d_dyn_scalars = (const uint32_t*)d_scalars;
  job.d_dyn_points = (const uint32_t*)d_points;
  job.d_dyn_offsets = (const uint64_t*)d_offsets;
  job.n_dyn = n_terms;
  job.n_msm = (uint32_t)batch;
  int rc = batch_device(c, job, accept_bitmap);
  if (rc != ZKGPU_OK) memset(accept_bitmap, 0, (batch + 7) / 8);
  return rc;
}

int zkgpu_verify_batch(zkgpu_ctx* c, const uint8_t* scalars, const uint8_t* points, const uint64_t* offsets,
                       size_t batch, uint8_t* accept_bitmap) {
  if (!c || !accept_bitmap || !offsets) return ZKGPU_EINVAL;
  memset(accept_bitmap, 0, (batch + 7) / 8);
  uint64_t n = 0;
  if (!offsets_ok(offsets, batch, &n) || (n && (!scalars || !points)) || batch >= (1ull << 31)) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  TRY(upload(c, c->in_scalars, scalars, n * 32));
  TRY(upload(c, c->in_points, points, n * 32));
  TRY(upload(c, c->in_offsets, offsets, (batch + 1) * 8));
  Job job;
  job.d_dyn_scalars = (const uint32_t*)c->in_scalars.p;
  job.d_dyn_points = (const uint32_t*)c->in_points.p;
  job.d_dyn_offsets = (const uint64_t*)c->in_offsets.p;
  job.n_dyn = n;
  job.n_msm = (uint32_t)batch;
  job.max_dyn_row = longest_row(offsets, batch);
  int rc = batch_device(c, job, accept_bitmap);
  if (rc != ZKGPU_OK) memset(accept_bitmap, 0, (batch + 7) / 8);
  return rc;
}

int zkgpu_pointset_create(zkgpu_ctx* c, const uint8_t* points, size_t n, zkgpu_pointset** out) {
  if (!c || !out || (n && !points) || n >= (1ull << 30)) return ZKGPU_EINVAL;
  *out = nullptr;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  TRY(upload(c, c->in_points, points, n * 32));
  TRY(ensure(c, c->status, 64));
  uint32_t* rows = nullptr;
  HIP_TRY(c, hipMalloc((void**)&rows, std::max<size_t>(n, 1) * NIELS_WORDS * 4));
  HIP_TRY(c, hipMemsetAsync(c->status.p, 0xff, 16, c->stream));
  unsigned long long* bad_index = (unsigned long long*)((char*)c->status.p + 8);
  if (n) {
    Launch l(c, "k_decompress");
    hipLaunchKernelGGL(k_decompress, dim3(blocks_for(n, 256)), dim3(256), 0, c->stream,
                       (const uint32_t*)c->in_points.p, rows, (uint64_t)n, (const uint64_t*)nullptr, 1u,
                       (uint32_t*)nullptr, bad_index, (uint8_t*)nullptr);
  }
  unsigned long long bad = ~0ull;
  hipError_t e = hipMemcpyAsync(&bad, bad_index, 8, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (c->profiling) prof_collect(c);
  if (e != hipSuccess) { (void)hipFree(rows); c->last_error = hipGetErrorString(e); return ZKGPU_EHIP; }
  if (bad != ~0ull) { (void)hipFree(rows); return ZKGPU_EINVALID_POINT; }
  zkgpu_pointset* ps = new zkgpu_pointset{c, rows, n};
  *out = ps;
  return ZKGPU_OK;
}

void zkgpu_pointset_destroy(zkgpu_pointset* ps) {
  if (!ps) return;
  DeviceGuard g(ps->ctx->device);
  (void)hipFree(ps->rows);
  if (ps->table) (void)hipFree(ps->table);
  delete ps;
}

size_t zkgpu_pointset_size(const zkgpu_pointset* ps) { return ps ? ps->n : 0; }

// bytes of the tables of n points at w bits, and of the scratch their construction needs beside them
static inline uint64_t table_bytes_for(uint64_t n, int w) { return (uint64_t)(255 / w + 1) * n * (1ull << (w - 1)) * TABLE_STRIDE * 4; }
static inline uint64_t table_build_bytes_for(uint64_t n, int w) {
  const uint64_t lanes = (uint64_t)(255 / w + 1) * n;
  return table_bytes_for(n, w) + lanes * (1ull << (w - 1)) * EXT_WORDS * 4 + lanes * EXT_WORDS * 4;
}

// window_bits == 0: the library chooses the KNEE, not the widest table that fits.  Wider windows mean fewer mixed additions
// per generator term (W = 255/w + 1 of them: 16 at 16 bits, 18 at 15, 19 at 14, 20 at 13) and exponentially more table
// (1026 generators: 25.9 / 14.6 / 7.7 / 4.0 GB).  Measured on MI355X (bench.py, setup.table_bits_sweep, rounds 3 and 4; device
// batches of 8192 distinct transactions): 16 / 15 / 14 / 13 bits = 3.32 / 3.31 / 3.30 / 3.20 M tx/s and 3.48 / 3.36 / 3.33 /
// 3.32 M -- the generator kernel is 8 % of the batch's instructions, so three more additions per term cost 0.5 - 4 %, and
// the widest table buys that with 18 GB.  Rule: take the widest width that is feasible (at most 16; tables no more than a
// quarter of the device's memory; construction -- tables + 1.7x scratch -- within 60 % of what is free right now), then the
// NARROWEST width whose addition count is within 19/16 of that one's ("the narrowest width within 3 % of the
// widest measured").  On a 288 GB MI355X: 14 bits (10.2 GB for 514 generators, 20.4 GB for 1026).  A caller who wants the
// last per cent passes the width itself (16).
int zkgpu_choose_table_bits(zkgpu_ctx* c, size_t n_points) {
  if (!c || n_points == 0) return ZKGPU_EINVAL;
  DeviceGuard g(c->device);
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return ZKGPU_EHIP;
  int widest = 4;
  for (int w = 16; w >= 4; --w)
    if (table_bytes_for(n_points, w) <= total_b / 4 && table_build_bytes_for(n_points, w) <= free_b / 10 * 6) { widest = w; break; }
  const int adds_widest = 255 / widest + 1;
  int knee = widest;
  for (int w = widest - 1; w >= 4; --w)
    if (16 * (255 / w + 1) <= 19 * adds_widest) knee = w; else break;
  return knee;
}

int zkgpu_pointset_table_bits(const zkgpu_pointset* ps) { return (ps && ps->table) ? ps->tbl_w : 0; }

int zkgpu_pointset_build_tables(zkgpu_ctx* c, zkgpu_pointset* ps, int window_bits) {
  if (!c || !ps || ps->ctx != c || ps->n == 0) return ZKGPU_EINVAL;
  if (window_bits == 0) {
    window_bits = zkgpu_choose_table_bits(c, ps->n);
    if (window_bits < 0) return window_bits;
  }
  if (window_bits < 2 || window_bits > 16) return ZKGPU_EINVAL;  // digits are int16
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  if (ps->table) { HIP_TRY(c, hipFree(ps->table)); ps->table = nullptr; }
  const int w = window_bits, W = 255 / w + 1;
  const uint32_t H = 1u << (w - 1);
  const uint64_t n_lanes = (uint64_t)W * ps->n;
  const uint64_t n_rows = n_lanes * H;
  uint32_t *base = nullptr, *tmp = nullptr, *table = nullptr;
  hipError_t e = hipMalloc((void**)&table, n_rows * TABLE_STRIDE * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&base, n_lanes * EXT_WORDS * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&tmp, n_rows * EXT_WORDS * 4);
  if (e != hipSuccess) {
    if (table) (void)hipFree(table);
    if (base) (void)hipFree(base);
    if (tmp) (void)hipFree(tmp);
    c->last_error = std::string("table allocation: ") + hipGetErrorString(e);
    return ZKGPU_ENOMEM;
  }
  {
    Launch l(c, "k_tbl_base");
    hipLaunchKernelGGL(k_tbl_base, dim3(blocks_for(ps->n, 64)), dim3(64), 0, c->stream, (const uint32_t*)ps->rows, base,
                       (uint32_t)ps->n, w, W);
  }
  {
    Launch l(c, "k_tbl_multiples");
    hipLaunchKernelGGL(k_tbl_multiples, dim3(blocks_for(n_lanes, 64)), dim3(64), 0, c->stream, (const uint32_t*)base,
                       tmp, table, n_lanes, H);
  }
  e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (c->profiling) prof_collect(c);
  (void)hipFree(base);
  (void)hipFree(tmp);
  if (e != hipSuccess) { (void)hipFree(table); c->last_error = hipGetErrorString(e); return ZKGPU_EHIP; }
  ps->table = table; ps->tbl_w = w; ps->tbl_W = W; ps->tbl_H = H;
  return ZKGPU_OK;
}

size_t zkgpu_pointset_table_bytes(const zkgpu_pointset* ps) {
  return (ps && ps->table) ? (size_t)ps->tbl_W * ps->n * ps->tbl_H * TABLE_STRIDE * 4 : 0;
}

// Test hook: the intermediate buffers of the last device-side preparation on this context
// ("challenges": n_ch_ext x 32 B per transaction, Montgomery form; "static_scalars", "dyn_scalars",
// "dyn_points": canonical 32-byte values).  Copies min(bytes, size) bytes; returns the bytes copied or < 0.
long long zkgpu_debug_read(zkgpu_ctx* c, const char* what, void* out, size_t bytes) {
  if (!c || !what || !out) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  const std::string w(what);
#ifdef ZK_PREP_STAMPS
  if (w == "prep_stamps") {
    const size_t n = std::min(bytes, sizeof(unsigned long long) * 256 * 2 * zk::PREP_STAMP_SLOTS);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(out, HIP_SYMBOL(zk::g_prep_stamps), n) != hipSuccess) return ZKGPU_EHIP;
    return (long long)n;
  }
#endif
  if (w == "prover_slices") {          // slices the LAST prover call on this context really ran in (run_sliced): 4 bytes
    if (bytes < 4) return ZKGPU_EINVAL;
    const uint32_t n = (uint32_t)c->last_prover_slices;
    memcpy(out, &n, 4);
    return 4;
  }
  const Buffer* b = w == "challenges" ? &c->prep_ch : w == "static_scalars" ? &c->prep_st_sc :
                    w == "dyn_scalars" ? &c->prep_dyn_sc : w == "dyn_points" ? &c->prep_dyn_pt : nullptr;
  if (!b || !b->p) return ZKGPU_EINVAL;
  const size_t n = std::min(bytes, b->cap);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, b->p, n, hipMemcpyDeviceToHost) != hipSuccess) return ZKGPU_EHIP;
  return (long long)n;
}

// Measurement aid: run the kernels of each batch one after another on a single stream, so that
// profiled durations are those of each kernel alone on the chip.
int zkgpu_set_serial(zkgpu_ctx* c, int on) {
  if (!c) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->serial = on != 0;
  return ZKGPU_OK;
}

// The provers (zkgpu_cloak_prove_batch, zkgpu_r1cs_prove_batch): 0 everything between the multiscalar multiplications on
// the device (prover_kernels.hpp), 1 host threads in lockstep (r1cs_prover.hpp).  Same proofs.
// mode 0 / 1 as the header says; 16 + S (S = 1 .. 8): the device prover with a call cut into S slices whatever its size
// (the tests prove nine statements in three slices; by default only calls of 1024 statements or more are sliced)
int zkgpu_set_prover_mode(zkgpu_ctx* c, int mode) {
  if (!c || mode < 0 || (mode > 1 && (mode < 17 || mode > 24))) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->prover_mode = mode > 1 ? 0 : mode;
  c->prover_slices = mode > 1 ? mode - 16 : 0;
  return ZKGPU_OK;
}

// Horner chains of a batch checked in groups: 0 automatic (by the share of failed groups in the last finished batch),
// 1 one chain per transaction up front, 2 one per group and then one per transaction of the failed groups
int zkgpu_set_horner_mode(zkgpu_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->horner_mode = mode;
  return ZKGPU_OK;
}

// Transcript replay: 0 automatic (one wavefront per transaction up to COOP_TRANSCRIPT_MAX transactions per batch,
// one lane per transaction beyond), 1 always one lane per transaction (k_transcript), 2 always one wavefront
// (k_tape_gather + k_transcript_coop + k_challenges).  Same results either way.
int zkgpu_set_transcript_mode(zkgpu_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->transcript_mode = mode;
  return ZKGPU_OK;
}

// Test hook for the cooperative Keccak (keccak_coop.hpp): `in` = 3 x 64 words (a, b, gather addresses), `out` =
// 8 x 64 words (row_ror:8, row_shr:1, row_shl:1 of a; both outputs of permlane16_swap(a, b) and of
// permlane32_swap(a, b); ds_bpermute(addr, a)); `states` = n_states x 25 u64, permuted in place by Keccak-f[1600]
// with one state per wavefront.
int zkgpu_debug_coop_selftest(zkgpu_ctx* c, const uint32_t* in, uint32_t* out, uint64_t* states, size_t n_states) {
  if (!c || !in || !out || (n_states && !states)) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  TRY(upload(c, c->in_scalars, in, 192 * 4));
  TRY(ensure(c, c->values, 512 * 4));
  TRY(upload(c, c->in_points, states, std::max<size_t>(n_states, 1) * 200));
  hipLaunchKernelGGL(k_coop_selftest, dim3((unsigned)std::max<size_t>(1, std::min<size_t>(n_states, 1024))), dim3(64), 0, c->stream,
                     (const uint32_t*)c->in_scalars.p, (uint32_t*)c->values.p, (uint2*)c->in_points.p, (uint32_t)n_states);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, c->values.p, 512 * 4, hipMemcpyDeviceToHost, c->stream));
  if (n_states) HIP_TRY(c, hipMemcpyAsync(states, c->in_points.p, n_states * 200, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ZKGPU_OK;
}

// Test hook: one arithmetic operation of the field / scalar layers on n elements (prover_kernels.hpp, k_debug_arith)
int zkgpu_debug_arith(zkgpu_ctx* c, int op, const uint8_t* a, const uint8_t* b, uint8_t* out, size_t n) {
  if (!c || !a || !b || !out || n == 0 || n >= (1u << 24)) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  TRY(upload(c, c->in_scalars, a, 32 * n));
  TRY(upload(c, c->in_points, b, 32 * n));
  TRY(ensure(c, c->values, 32 * n));
  hipLaunchKernelGGL(k_debug_arith, dim3(blocks_for(n, 64)), dim3(64), 0, c->stream, (uint32_t)op, (const uint32_t*)c->in_scalars.p,
                     (const uint32_t*)c->in_points.p, (uint32_t*)c->values.p, (uint32_t)n);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, c->values.p, 32 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return ZKGPU_OK;
}

// Failed groups: 0 automatic (locate the culprit from LOCATE_MIN_BATCH transactions per batch on), 1 always re-check
// every transaction of a failed group, 2 always locate.  Same verdicts either way.
int zkgpu_set_locate_mode(zkgpu_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 3) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->locate_mode = mode;
  return ZKGPU_OK;
}

// Test hook: with on != 0 every located transaction is treated as NOT accounting for its group's sum, which sends
// the batch down the ungrouped re-run in zkgpu_verify_wait; returns how many such re-runs this context has made.
long long zkgpu_debug_force_regroup(zkgpu_ctx* c, int on) {
  if (!c) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->force_unresolved = on != 0;
  return (long long)c->regroup_fallbacks;
}

int zkgpu_set_group_size(zkgpu_ctx* c, int group) {
  if (!c || group < 1 || group > 64) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->group_size = group;
  return ZKGPU_OK;
}

int zkgpu_set_static_parts(zkgpu_ctx* c, int parts) {
  if (!c || parts < 0 || parts > 64) return ZKGPU_EINVAL;
  c->forced_parts = parts;
  return ZKGPU_OK;
}

int zkgpu_set_tail_mode(zkgpu_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 1) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->tail_mode = mode;
  return ZKGPU_OK;
}

int zkgpu_set_locate_parts(zkgpu_ctx* c, int parts) {
  if (!c || parts < 0 || parts > 64) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->locate_parts = parts;
  return ZKGPU_OK;
}

int zkgpu_verify_batch_ps_dev(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t batch, const void* d_dyn_scalars,
                              const void* d_dyn_points, const void* d_dyn_offsets, size_t n_dyn,
                              const void* d_static_scalars, const void* d_static_index,
                              const void* d_static_offsets, size_t n_static, uint8_t* accept_bitmap) {
  // a point set may be used from any context of the same device (its rows and tables are read-only)
  if (!c || !ps || ps->ctx->device != c->device || !accept_bitmap || batch >= (1ull << 31)) return ZKGPU_EINVAL;
  if (batch && (!d_dyn_offsets || !d_static_offsets)) return ZKGPU_EINVAL;
  if ((n_dyn && (!d_dyn_scalars || !d_dyn_points)) || (n_static && !d_static_scalars)) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  memset(accept_bitmap, 0, (batch + 7) / 8);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  Job job;
  job.d_dyn_scalars = (const uint32_t*)d_dyn_scalars;
  job.d_dyn_points = (const uint32_t*)d_dyn_points;
  job.d_dyn_offsets = (const uint64_t*)d_dyn_offsets;
  job.n_dyn = n_dyn;
  job.d_st_scalars = (const uint32_t*)d_static_scalars;
  job.d_st_index = (const uint32_t*)d_static_index;
  job.d_st_offsets = (const uint64_t*)d_static_offsets;
  job.n_static = n_static;
  job.d_static_rows = ps->rows;
  job.n_msm = (uint32_t)batch;
  int rc = (ps->table && n_static) ? batch_device_tables(c, job, ps, accept_bitmap) : batch_device(c, job, accept_bitmap);
  if (rc != ZKGPU_OK) memset(accept_bitmap, 0, (batch + 7) / 8);
  return rc;
}

int zkgpu_verify_batch_ps(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t batch, const uint8_t* dyn_scalars,
                          const uint8_t* dyn_points, const uint64_t* dyn_offsets, const uint8_t* static_scalars,
                          const uint32_t* static_index, const uint64_t* static_offsets, uint8_t* accept_bitmap) {
  if (!c || !ps || ps->ctx->device != c->device || !accept_bitmap || !dyn_offsets || !static_offsets) return ZKGPU_EINVAL;
  memset(accept_bitmap, 0, (batch + 7) / 8);
  uint64_t nd = 0, ns = 0;
  if (!offsets_ok(dyn_offsets, batch, &nd) || !offsets_ok(static_offsets, batch, &ns)) return ZKGPU_EINVAL;
  if ((nd && (!dyn_scalars || !dyn_points)) || (ns && !static_scalars)) return ZKGPU_EINVAL;
  // static indices must name points of the set
  if (static_index) {
    for (uint64_t k = 0; k < ns; ++k) if (static_index[k] >= ps->n) return ZKGPU_EINVAL;
  } else {
    for (size_t i = 0; i < batch; ++i) if (static_offsets[i + 1] - static_offsets[i] > ps->n) return ZKGPU_EINVAL;
  }
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  {
    DeviceGuard g(c->device);
    TRY(upload(c, c->in_scalars, dyn_scalars, nd * 32));
    TRY(upload(c, c->in_points, dyn_points, nd * 32));
    TRY(upload(c, c->in_offsets, dyn_offsets, (batch + 1) * 8));
    TRY(upload(c, c->in_st_scalars, static_scalars, ns * 32));
    if (static_index) TRY(upload(c, c->in_st_index, static_index, ns * 4));
    TRY(upload(c, c->in_st_offsets, static_offsets, (batch + 1) * 8));
  }
  return zkgpu_verify_batch_ps_dev(c, ps, batch, c->in_scalars.p, c->in_points.p, c->in_offsets.p, nd,
                                   c->in_st_scalars.p, static_index ? c->in_st_index.p : nullptr,
                                   c->in_st_offsets.p, ns, accept_bitmap);
}

namespace {
// values of `batch` multiscalar multiplications over the tables, terms resident on the device
// (d_scalars: n x 8 canonical words, d_index: n generator indices or nullptr, d_offsets: batch + 1) -> 32 bytes each on the host
int msm_ps_core(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t batch, uint64_t n, const uint32_t* d_scalars, const uint32_t* d_index,
                const uint64_t* d_offsets, uint8_t* out) {
  hipStream_t s = c->stream;
  const int W = ps->tbl_W;
  const int P = (int)std::max<uint64_t>(1, std::min<uint64_t>(64, (131072 + batch * W - 1) / (batch * W)));
  const uint64_t n_lanes = (uint64_t)batch * W * P;
  TRY(ensure(c, c->status, 64));
  TRY(ensure(c, c->digits, std::max<uint64_t>(n, 1) * W * 2));
  TRY(ensure(c, c->st_partials, n_lanes * EXT_WORDS * 4));
  TRY(ensure(c, c->values, 32 * batch));
  TRY(ensure_pinned(c, 32 * batch + 64));
  HIP_TRY(c, hipMemsetAsync(c->status.p, 0, 8, s));
  if (n) {
    Launch l(c, "k_static_digits");
    hipLaunchKernelGGL(k_static_digits, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_scalars,
                       (int16_t*)c->digits.p, n, ps->tbl_w, W, (uint32_t*)c->status.p, (const uint32_t*)nullptr,
                       (const uint32_t*)nullptr, 0u, 0u);
  }
  {
    Launch l(c, "k_static_accumulate");
    hipLaunchKernelGGL(k_static_accumulate<false>, dim3(blocks_for(n_lanes, 256)), dim3(256), 0, s, (const int16_t*)c->digits.p,
                       d_offsets, d_index, (const uint32_t*)ps->table, (uint32_t)ps->n, ps->tbl_H, W, P, (uint32_t)batch, n,
                       (uint32_t*)c->st_partials.p, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 1u);
  }
  {
    Launch l(c, "k_static_values");
    hipLaunchKernelGGL(k_static_values, dim3((unsigned)batch), dim3(64), 0, s, (const uint32_t*)c->st_partials.p,
                       (uint32_t)(W * P), (uint32_t*)c->values.p);
  }
  HIP_TRY(c, hipGetLastError());
  char* h = (char*)c->pinned;
  HIP_TRY(c, hipMemcpyAsync(h, c->values.p, 32 * batch, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(h + 32 * batch, c->status.p, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  if (c->profiling) prof_collect(c);
  uint32_t st;
  memcpy(&st, h + 32 * batch, 4);
  if (st & 2u) { c->last_error = "scalar with bit 255 set"; return ZKGPU_EINVAL; }
  memcpy(out, h, 32 * batch);
  return ZKGPU_OK;
}
}  // namespace

// Values of `batch` multiscalar multiplications over the points of a resident set that carries
// fixed-base tables: out[32 m] = ENCODE(sum_k scalars[k] * ps[index[k]]), k in [offsets[m], offsets[m+1]).
// This is the prover's primitive (Pedersen vector commitments A_I, A_O, S, T_i, and every L_j / R_j of an
// inner-product argument kept as coefficient vectors over the original generators): one mixed
// addition per term and window, no doublings.  index == NULL: term k of row m uses point k - offsets[m].
int zkgpu_msm_ps_batch(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t batch, const uint8_t* scalars,
                       const uint32_t* index, const uint64_t* offsets, uint8_t* out) {
  if (!c || !ps || ps->ctx->device != c->device || !offsets || !out || batch >= (1ull << 31)) return ZKGPU_EINVAL;
  memset(out, 0, 32 * batch);
  if (batch == 0) return ZKGPU_OK;
  if (!ps->table) { c->last_error = "zkgpu_msm_ps_batch needs zkgpu_pointset_build_tables first"; return ZKGPU_EINVAL; }
  uint64_t n = 0;
  if (!offsets_ok(offsets, batch, &n) || (n && !scalars)) return ZKGPU_EINVAL;
  if (index) {
    for (uint64_t k = 0; k < n; ++k) if (index[k] >= ps->n) return ZKGPU_EINVAL;
  } else {
    for (size_t i = 0; i < batch; ++i) if (offsets[i + 1] - offsets[i] > ps->n) return ZKGPU_EINVAL;
  }
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  TRY(upload(c, c->in_st_scalars, scalars, n * 32));
  if (index) TRY(upload(c, c->in_st_index, index, n * 4));
  TRY(upload(c, c->in_st_offsets, offsets, (batch + 1) * 8));
  return msm_ps_core(c, ps, batch, n, (const uint32_t*)c->in_st_scalars.p, index ? (const uint32_t*)c->in_st_index.p : nullptr,
                     (const uint64_t*)c->in_st_offsets.p, out);
}

// Proves `batch` cloak statements of one shape (BASELINE.json configs[4]).
// The provers run in lockstep on host threads (r1cs_prover.hpp: transcripts, witness, polynomial and
// inner-product algebra); every phase of the whole batch is ONE zkgpu_msm_ps_batch over the generator
// tables of `ps` = [B, B_blinding, G_0..G_{cap-1}, H_0..H_{cap-1}]: 2 (n_in + n_out) value commitments,
// A_I A_O S of both phases, T_1 T_3..T_6, then L_j R_j of each inner-product round (513 terms each
// for padded n = 256).  quantities: batch x (n_in + n_out) u64; flavors: 32 B each; seeds: 32 B per
// statement (blindings and TranscriptRng randomness are derived from it).  commitments: batch x
// 64 (n_in + n_out) B; proofs: batch x proof_stride B, *proof_len bytes used of each.
namespace {
bool desc_from_c(zkgpu_ctx* c, const zkgpu_r1cs_desc* d, R1csDesc& desc);

int ipa_on_device(zkgpu_ctx* c, const zkgpu_pointset* ps, std::vector<std::unique_ptr<R1csProver>>& pr,
                  const std::function<void(const std::function<void(size_t)>&)>& parallel) {
  const size_t batch = pr.size();
  const size_t pn = pr[0]->ipa_len(), k = pr[0]->ipa_rounds(), cap = pr[0]->gens_capacity();
  for (size_t i = 0; i < batch; ++i)
    if (!pr[i]->at_ipa() || pr[i]->ipa_len() != pn) { c->last_error = "prover: statements of one batch must share one shape"; return ZKGPU_EINVAL; }
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  hipStream_t s = c->stream;
  const size_t vec = pn * 32, row_len = pn + 1, n_rows = 2 * batch;
  std::vector<uint8_t> h_lv(batch * vec), h_rv(batch * vec), h_cg(batch * vec), h_ch(batch * vec), h_w(batch * 32);
  parallel([&](size_t i) { pr[i]->ipa_export(&h_lv[i * vec], &h_rv[i * vec], &h_cg[i * vec], &h_ch[i * vec], &h_w[i * 32]); });
  TRY(upload(c, c->ipa_lv, h_lv.data(), h_lv.size()));
  TRY(upload(c, c->ipa_rv, h_rv.data(), h_rv.size()));
  TRY(upload(c, c->ipa_cg, h_cg.data(), h_cg.size()));
  TRY(upload(c, c->ipa_ch, h_ch.data(), h_ch.size()));
  TRY(upload(c, c->ipa_w, h_w.data(), h_w.size()));
  TRY(ensure(c, c->ipa_u, batch * 64));
  TRY(ensure(c, c->values, 64 * batch));
  TRY(ensure(c, c->in_st_scalars, n_rows * row_len * 32));
  TRY(ensure(c, c->in_st_index, n_rows * row_len * 4));
  std::vector<uint64_t> offs(n_rows + 1);
  for (size_t r = 0; r <= n_rows; ++r) offs[r] = r * row_len;
  TRY(upload(c, c->in_st_offsets, offs.data(), offs.size() * 8));
  hipLaunchKernelGGL(k_ipa_to_mont, dim3(blocks_for(batch * pn, 256)), dim3(256), 0, s, (uint32_t*)c->ipa_lv.p, (uint64_t)batch * pn);
  hipLaunchKernelGGL(k_ipa_to_mont, dim3(blocks_for(batch * pn, 256)), dim3(256), 0, s, (uint32_t*)c->ipa_rv.p, (uint64_t)batch * pn);
  std::vector<uint8_t> pts(32 * n_rows), uu(64 * batch), ab(64 * batch);
  size_t len = pn;
  for (size_t round = 0; round <= k; ++round) {
    const bool last = round == k;
    {
      Launch l(c, "k_ipa_round");
      hipLaunchKernelGGL(k_ipa_round, dim3((unsigned)batch), dim3(256), 0, s, (uint32_t*)c->ipa_lv.p, (uint32_t*)c->ipa_rv.p,
                         (uint32_t*)c->ipa_cg.p, (uint32_t*)c->ipa_ch.p, (const uint32_t*)c->ipa_w.p, (const uint32_t*)c->ipa_u.p,
                         (uint32_t)pn, (uint32_t)len, (uint32_t)cap, round > 0 ? 1u : 0u, last ? 0u : 1u,
                         (uint32_t*)c->in_st_scalars.p, (uint32_t*)c->in_st_index.p, (uint32_t*)c->values.p);
    }
    HIP_TRY(c, hipGetLastError());
    if (last) break;
    TRY(msm_ps_core(c, ps, n_rows, (uint64_t)n_rows * row_len, (const uint32_t*)c->in_st_scalars.p, (const uint32_t*)c->in_st_index.p,
                    (const uint64_t*)c->in_st_offsets.p, pts.data()));
    parallel([&](size_t i) { pr[i]->ipa_absorb(&pts[64 * i], &uu[64 * i]); });     // L, R in; u, 1/u out
    HIP_TRY(c, hipMemcpyAsync(c->ipa_u.p, uu.data(), uu.size(), hipMemcpyHostToDevice, s));
    len /= 2;
  }
  HIP_TRY(c, hipMemcpyAsync(ab.data(), c->values.p, 64 * batch, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  if (c->profiling) prof_collect(c);
  parallel([&](size_t i) { pr[i]->ipa_finish(&ab[64 * i], &ab[64 * i + 32]); });
  return ZKGPU_OK;
}

// ---- the device prover ------------------------------------------------------------------------------------
// (host_parallel: host_pool.hpp)
#ifndef PV_RNG_LANES_FROM
#define PV_RNG_LANES_FROM ((size_t)1 << 40)      // (never, until the A/B of round 6 says otherwise)
#endif

// values of `rows` multiscalar multiplications over the tables, everything resident: queued on the context's stream,
// encodings written to d_out (32 bytes per row)
// kinds: 1, or the rows come in groups of `kinds` rows of different make per proof (A_I, A_O, S): k_static_accumulate then puts
// rows of one make side by side in its wavefronts, where their zero digits coincide and the additions nobody needs are skipped
int msm_ps_dev(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t rows, uint64_t n, const uint32_t* d_scalars, const uint32_t* d_index,
               const uint64_t* d_offsets, uint32_t* d_out, uint32_t kinds = 1) {
  hipStream_t s = c->stream;
  const int W = ps->tbl_W;
  const int P = (int)std::max<uint64_t>(1, std::min<uint64_t>(64, (131072 + rows * W - 1) / (rows * W)));
  const uint64_t n_lanes = (uint64_t)rows * W * P;
  TRY(ensure(c, c->status, 64));
  TRY(ensure(c, c->digits, std::max<uint64_t>(n, 1) * W * 2));
  TRY(ensure(c, c->st_partials, n_lanes * EXT_WORDS * 4));
  if (n) {
    Launch l(c, "k_static_digits");
    hipLaunchKernelGGL(k_static_digits, dim3(blocks_for(n, 256)), dim3(256), 0, s, d_scalars, (int16_t*)c->digits.p, n, ps->tbl_w, W,
                       (uint32_t*)c->status.p, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, 1u);     // (results are encoded: small negatives folded)
  }
  {
    Launch l(c, "k_static_accumulate");
    hipLaunchKernelGGL(k_static_accumulate<true>, dim3(blocks_for(n_lanes, 256)), dim3(256), 0, s, (const int16_t*)c->digits.p, d_offsets,
                       d_index, (const uint32_t*)ps->table, (uint32_t)ps->n, ps->tbl_H, W, P, (uint32_t)rows, n,
                       (uint32_t*)c->st_partials.p, (const uint32_t*)nullptr, (const uint32_t*)nullptr, rows % kinds == 0 ? kinds : 1u);
  }
  TRY(ensure(c, c->rechk_pts, rows * EXT_WORDS * 4));       // (free while a prover runs: the verifier's re-check sums)
  {
    Launch l(c, "k_static_row_sums");
    hipLaunchKernelGGL(k_static_row_sums, dim3(blocks_for(rows * ROW_SUM_LANES, 64)), dim3(64), 0, s, (const uint32_t*)c->st_partials.p,
                       (uint32_t)(W * P), (uint32_t)rows, (uint32_t*)c->rechk_pts.p);
  }
  {
    Launch l(c, "k_encode_rows");
    hipLaunchKernelGGL(k_encode_rows, dim3(blocks_for(rows, 64)), dim3(64), 0, s, (const uint32_t*)c->rechk_pts.p, (uint32_t)rows, d_out);
  }
  HIP_TRY(c, hipGetLastError());
  return ZKGPU_OK;
}

// The TranscriptRng's draws -- one Keccak-f each, 3 + 2n of them per proof, a third of a proof's instructions in the wavefront
// form (k_pv_rng_coop: 25 of 64 lanes at work, ~1500 wave-instructions per draw) -- in the LANE form (k_pv_rng: 64 proofs per
// wavefront, ~70 wave-instructions per draw and proof, but 8 us of one lane's latency per draw) when the call is large enough
// for other slices to fill the chip meanwhile.  Same draws, same proofs.  (ZKGPU_PV_RNG=lanes|coop overrides: A/B.)
bool pv_rng_lanes_pay(const zkgpu_ctx* c, size_t batch_of_this_slice) {
  (void)c;
  return batch_of_this_slice >= PV_RNG_LANES_FROM;
}

// Proves `batch` statements of one described system.  Host: the constant tables (once per call), the inputs' upload,
// a queue of kernels; device: everything else.  values / blindings: batch x m x 32 bytes (any 256-bit values, reduced
// on the device), given: batch x n_given x 64, rng_seeds: batch x 32.
int prove_device(zkgpu_ctx* c, const zkgpu_pointset* ps, const PvHostPlan& hp, size_t batch, const uint8_t* values, const uint8_t* blindings,
                 const uint8_t* given, const uint8_t* rng_seeds, uint8_t* commitments, uint8_t* proofs, size_t proof_stride, size_t* proof_len) {
  const PvShape& sh = hp.sh;
  if (sh.proof_len > proof_stride) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  hipStream_t s = c->stream;
  const bool timing = getenv("ZKGPU_PROVER_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_start = now();
  // constant tables: one blob
  std::vector<uint32_t> blob;
  auto put = [&blob](const std::vector<uint32_t>& v) { const size_t at = blob.size(); blob.insert(blob.end(), v.begin(), v.end()); while (blob.size() & 3) blob.push_back(0); return at; };
  const size_t o_init = put(hp.init), o_mc = put(hp.mono_chal), o_mp = put(hp.mono_pow), o_co = put(hp.con_off), o_tk = put(hp.t_kind),
               o_ti = put(hp.t_idx), o_tm = put(hp.t_mono), o_tc = put(hp.t_coef), o_md = put(hp.mult_def), o_gs = put(hp.given_slot),
               o_to = put(hp.tgt_off), o_tinf = put(hp.term_info), o_pq = put(hp.prod_qm), o_pc = put(hp.prod_coef);
  std::vector<uint32_t> labels((hp.chal_labels.size() + 3) / 4, 0);
  memcpy(labels.data(), hp.chal_labels.data(), hp.chal_labels.size());
  const size_t o_lab = put(labels);
  (void)put({sh.m, sh.n1, sh.n, sh.pn, sh.gens_capacity});   // part of the key the cached scaffolding goes by
  if (blob != c->pv_plan_host) {        // the same statement as in the last call: the tables are already there
    TRY(upload(c, c->pv_plan, blob.data(), blob.size() * 4));
    c->pv_plan_host = blob;
    c->pv_lay_batch = 0;
  }
  const uint32_t* base = (const uint32_t*)c->pv_plan.p;
  PvPlan P;
  P.init = base + o_init; P.chal_labels = (const uint8_t*)(base + o_lab); P.mono_chal = base + o_mc; P.mono_pow = base + o_mp;
  P.con_off = base + o_co; P.t_kind = base + o_tk; P.t_idx = base + o_ti; P.t_mono = base + o_tm; P.t_coef = base + o_tc;
  P.mult_def = base + o_md; P.given_slot = base + o_gs; P.tgt_off = base + o_to; P.term_info = base + o_tinf; P.prod_qm = base + o_pq;
  P.prod_coef = base + o_pc;
  // inputs: values | blindings | given | seeds
  const size_t b_val = batch * sh.m * 32, b_giv = batch * (size_t)sh.n_given * 64, b_seed = batch * 32;
  TRY(ensure(c, c->pv_in, 2 * b_val + b_giv + b_seed + 64));
  char* in = (char*)c->pv_in.p;
  if (b_val) {
    HIP_TRY(c, hipMemcpyAsync(in, values, b_val, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(in + b_val, blindings, b_val, hipMemcpyHostToDevice, s));
  }
  if (b_giv) HIP_TRY(c, hipMemcpyAsync(in + 2 * b_val, given, b_giv, hipMemcpyHostToDevice, s));
  HIP_TRY(c, hipMemcpyAsync(in + 2 * b_val + b_giv, rng_seeds, b_seed, hipMemcpyHostToDevice, s));
  // rows and their scaffolding (offsets and generator indices: kept on the device from call to call while statement and
  // batch size stay the same)
  const size_t cap = sh.gens_capacity;
  const size_t row_len = (size_t)sh.pn + 1, n_ipa_rows = 2 * batch;
  const size_t lay_rows[4] = {batch * sh.m, 3 * batch, 3 * batch, 5 * batch};
  const size_t lay_terms[4] = {2 * batch * sh.m, batch * (size_t)sh.r1_terms, batch * (size_t)sh.r2_terms, 10 * batch};
  size_t o_off[4], o_idx[4], o_ipa, lay_bytes = 0;
  auto place = [&lay_bytes](size_t bytes) { const size_t at = lay_bytes; lay_bytes = (lay_bytes + bytes + 15) & ~(size_t)15; return at; };
  for (int i = 0; i < 4; ++i) { o_off[i] = place((lay_rows[i] + 1) * 8); o_idx[i] = place(lay_terms[i] * 4); }
  o_ipa = place((n_ipa_rows + 1) * 8);
  if (c->pv_lay_batch != batch) {
    const PvRows l0 = pv_rows_pairs(batch * sh.m), l1 = pv_rows_commit(batch, 0, sh.n1, cap, false),
                 l2 = pv_rows_commit(batch, sh.n1, sh.n, cap, true), l3 = pv_rows_pairs(batch * 5);
    const PvRows* lays[4] = {&l0, &l1, &l2, &l3};
    std::vector<uint8_t> lay(lay_bytes, 0);
    for (int i = 0; i < 4; ++i) {
      if (lays[i]->offsets.size() != lay_rows[i] + 1 || lays[i]->index.size() != lay_terms[i]) { c->last_error = "prover: row scaffolding out of step"; return ZKGPU_EINVAL; }
      memcpy(&lay[o_off[i]], lays[i]->offsets.data(), lays[i]->offsets.size() * 8);
      memcpy(&lay[o_idx[i]], lays[i]->index.data(), lays[i]->index.size() * 4);
    }
    for (size_t r = 0; r <= n_ipa_rows; ++r) { const uint64_t v = r * row_len; memcpy(&lay[o_ipa + 8 * r], &v, 8); }
    TRY(upload(c, c->pv_lay, lay.data(), lay.size()));
    c->pv_lay_batch = batch;
  }
  const char* lb = (const char*)c->pv_lay.p;
  TRY(ensure(c, c->pv_state, batch * (size_t)sh.state_words * 4));
  TRY(ensure(c, c->pv_rows0, std::max<size_t>(batch * sh.m, 1) * 64));
  TRY(ensure(c, c->pv_rows1, batch * (size_t)sh.r1_terms * 32));
  TRY(ensure(c, c->pv_rows2, std::max<size_t>(batch * (size_t)sh.r2_terms, 1) * 32));
  TRY(ensure(c, c->pv_rows3, batch * 5 * 64));
  TRY(ensure(c, c->pv_pts, std::max<size_t>(batch * 5, batch * sh.m) * 32 + 64));
  TRY(ensure(c, c->pv_com, std::max<size_t>(batch * sh.m, 1) * 32));
  TRY(ensure(c, c->pv_ab, batch * 64));
  TRY(ensure(c, c->pv_proofs, batch * (size_t)sh.proof_stride));
  const size_t vec = (size_t)sh.pn * 32;
  TRY(ensure(c, c->ipa_lv, batch * vec)); TRY(ensure(c, c->ipa_rv, batch * vec)); TRY(ensure(c, c->ipa_cg, batch * vec));
  TRY(ensure(c, c->ipa_ch, batch * vec)); TRY(ensure(c, c->ipa_w, batch * 32)); TRY(ensure(c, c->ipa_u, batch * 64));
  TRY(ensure(c, c->in_st_scalars, n_ipa_rows * row_len * 32));
  TRY(ensure(c, c->in_st_index, n_ipa_rows * row_len * 4));
  TRY(ensure(c, c->status, 64));
  HIP_TRY(c, hipMemsetAsync(c->status.p, 0, 64, s));
  PvBatch B;
  B.state = (uint32_t*)c->pv_state.p;
  B.values = (const uint32_t*)in; B.blindings = (const uint32_t*)(in + b_val); B.given = (const uint32_t*)(in + 2 * b_val);
  B.rng_seed = (const uint32_t*)(in + 2 * b_val + b_giv);
  B.proofs = (uint8_t*)c->pv_proofs.p;
  B.rows0 = (uint32_t*)c->pv_rows0.p; B.rows1 = (uint32_t*)c->pv_rows1.p; B.rows2 = (uint32_t*)c->pv_rows2.p; B.rows3 = (uint32_t*)c->pv_rows3.p;
  B.ipa_lv = (uint32_t*)c->ipa_lv.p; B.ipa_rv = (uint32_t*)c->ipa_rv.p; B.ipa_cg = (uint32_t*)c->ipa_cg.p; B.ipa_ch = (uint32_t*)c->ipa_ch.p;
  B.ipa_w = (uint32_t*)c->ipa_w.p; B.ipa_u = (uint32_t*)c->ipa_u.p;
  uint32_t* pts = (uint32_t*)c->pv_pts.p;
  auto msm = [&](int which, const uint32_t* rows, uint32_t* out) {
    return msm_ps_dev(c, ps, lay_rows[which], lay_terms[which], rows, (const uint32_t*)(lb + o_idx[which]),
                      (const uint64_t*)(lb + o_off[which]), out, which == 1 || which == 2 ? 3u : 1u);
  };
  const unsigned nb = (unsigned)batch;
  const double t_setup = now();
  { Launch l(c, "k_pv_phase0"); hipLaunchKernelGGL(k_pv_phase0, dim3(nb), dim3(64), 0, s, sh, B); }
  if (sh.m) TRY(msm(0, B.rows0, (uint32_t*)c->pv_com.p));
  // a phase = its stages, each a launch: one lane per proof where a stage is one thread's work, a workgroup per proof elsewhere
  const unsigned nbl = blocks_for(nb, 64);
#define PV_LANES(PH, ST, pts_) { Launch l(c, "k_pv_lanes_" #PH "_" #ST); hipLaunchKernelGGL((k_pv_lanes<PH, ST>), dim3(nbl), dim3(64), 0, s, sh, P, B, (const uint32_t*)(pts_), nb); }
#define PV_WG(PH, ST, pts_) { Launch l(c, "k_pv_wg_" #PH "_" #ST); hipLaunchKernelGGL((k_pv_wg<PH, ST>), dim3(nb), dim3(256), 0, s, sh, P, B, (const uint32_t*)(pts_)); }
  PV_LANES(1, 1, c->pv_com.p) PV_WG(1, 2, c->pv_com.p) PV_LANES(1, 4, c->pv_com.p) PV_WG(1, 8, c->pv_com.p)
  // the TranscriptRng's draws: one wavefront per proof on the spread Keccak state (k_pv_rng, one lane per proof, is the
  // form the host emulation and the first device version ran; kept for comparison)
  static const int rng_env = [] { const char* e = getenv("ZKGPU_PV_RNG"); return e ? (e[0] == 'l' ? 1 : e[0] == 'c' ? 2 : 0) : 0; }();
  const bool rng_lanes = rng_env == 1 || (rng_env == 0 && pv_rng_lanes_pay(c, batch));
  auto rng_launch = [&](uint32_t phase, uint32_t) {
    if (rng_lanes) {
      Launch l(c, "k_pv_rng");
      hipLaunchKernelGGL(k_pv_rng, dim3(blocks_for(nb, 64)), dim3(64), 0, s, sh, B, nb, phase);
    } else {
      Launch l(c, "k_pv_rng_coop");
      hipLaunchKernelGGL(k_pv_rng_coop, dim3(nb), dim3(64), 0, s, sh, B, nb, phase);
    }
  };
  rng_launch(1u, sh.n1);
  TRY(msm(1, B.rows1, pts));
  PV_LANES(2, 1, pts)
  if (sh.n > sh.n1) { PV_WG(2, 2, pts) PV_LANES(2, 4, pts) PV_WG(2, 8, pts) }
  if (sh.n > sh.n1) rng_launch(2u, sh.n - sh.n1);
  TRY(msm(2, B.rows2, pts));
  PV_LANES(3, 1, pts) PV_WG(3, 2, pts) PV_LANES(3, 4, pts)
  TRY(msm(3, B.rows3, pts));
  PV_LANES(4, 1, pts) PV_WG(4, 2, pts)
#undef PV_LANES
#undef PV_WG
  size_t len = sh.pn;
  for (uint32_t round = 0; round <= sh.k; ++round) {
    const bool last = round == sh.k;
    {
      Launch l(c, "k_ipa_round");
      hipLaunchKernelGGL(k_ipa_round, dim3(nb), dim3(256), 0, s, B.ipa_lv, B.ipa_rv, B.ipa_cg, B.ipa_ch, (const uint32_t*)B.ipa_w,
                         (const uint32_t*)B.ipa_u, sh.pn, (uint32_t)len, (uint32_t)cap, round > 0 ? 1u : 0u, last ? 0u : 1u,
                         (uint32_t*)c->in_st_scalars.p, (uint32_t*)c->in_st_index.p, (uint32_t*)c->pv_ab.p);
    }
    if (last) break;
    TRY(msm_ps_dev(c, ps, n_ipa_rows, (uint64_t)n_ipa_rows * row_len, (const uint32_t*)c->in_st_scalars.p, (const uint32_t*)c->in_st_index.p,
                   (const uint64_t*)(lb + o_ipa), pts));
    { Launch l(c, "k_pv_ipa_lanes"); hipLaunchKernelGGL(k_pv_ipa_lanes, dim3(blocks_for(nb, 64)), dim3(64), 0, s, sh, B, round, (const uint32_t*)pts, nb); }
    len /= 2;
  }
  { Launch l(c, "k_pv_finish"); hipLaunchKernelGGL(k_pv_finish, dim3(blocks_for(batch, 256)), dim3(256), 0, s, sh, B, (const uint32_t*)c->pv_ab.p, nb, (uint32_t*)c->status.p); }
  HIP_TRY(c, hipGetLastError());
  std::vector<uint8_t> h_proofs(batch * (size_t)sh.proof_stride);
  uint32_t st[2] = {0, 0};
  HIP_TRY(c, hipMemcpyAsync(h_proofs.data(), c->pv_proofs.p, h_proofs.size(), hipMemcpyDeviceToHost, s));
  if (sh.m) HIP_TRY(c, hipMemcpyAsync(commitments, c->pv_com.p, batch * sh.m * 32, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(st, c->status.p, 8, hipMemcpyDeviceToHost, s));
  const double t_queued = now();
  HIP_TRY(c, hipStreamSynchronize(s));
  if (timing) fprintf(stderr, "device prover: setup + uploads %.1f ms, queueing %.1f ms, waiting for the device %.1f ms (%zu proofs)\n",
                      (t_setup - t_start) * 1e3, (t_queued - t_setup) * 1e3, (now() - t_queued) * 1e3, batch);
  if (c->profiling) prof_collect(c);
  if (st[0] & 1u) { c->last_error = "prover: inconsistent witness (a multiplier's defining constraint cannot be solved)"; return ZKGPU_EINVAL; }
  if (st[0] & 2u) { c->last_error = "prover: scalar out of range or TranscriptRng out of step"; return ZKGPU_EHIP; }
  for (size_t i = 0; i < batch; ++i) memcpy(proofs + proof_stride * i, h_proofs.data() + (size_t)sh.proof_stride * i, sh.proof_len);
  *proof_len = sh.proof_len;
  return ZKGPU_OK;
}

// Runs `batch` provers in lockstep: every phase of the whole batch is ONE zkgpu_msm_ps_batch over the tables.
int prove_lockstep(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t batch, int host_threads,
                   const std::function<std::unique_ptr<R1csProver>(size_t)>& make, uint8_t* commitments, size_t com_bytes,
                   uint8_t* proofs, size_t proof_stride, size_t* proof_len) {
  std::vector<std::unique_ptr<R1csProver>> pr(batch);
  std::vector<std::vector<MsmRow>> rows(batch);
  const int nt = std::max(1, std::min<int>(host_threads > 0 ? host_threads : usable_cpus(), 256));
  auto parallel = [&](const std::function<void(size_t)>& f) {
    if (nt == 1 || batch == 1) { for (size_t i = 0; i < batch; ++i) f(i); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { for (size_t i = (size_t)t; i < batch; i += (size_t)nt) f(i); });
    for (auto& t : th) t.join();
  };
  parallel([&](size_t i) {
    pr[i] = make(i);
    pr[i]->set_device_ipa(true);
    pr[i]->begin(rows[i]);
  });
  std::vector<uint64_t> offs, row_of(batch + 1);
  std::vector<uint8_t> sc, pts;
  std::vector<uint32_t> idx;
  const bool timing = getenv("ZKGPU_PROVER_TIMING") != nullptr;
  double t_flat = 0, t_gpu = 0, t_step = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (int phase = 0;; ++phase) {
    bool any = false, bad = false;
    for (size_t i = 0; i < batch; ++i) { any |= !pr[i]->done(); bad |= pr[i]->failed(); }
    if (bad) { c->last_error = "prover: inconsistent witness or too few generators"; return ZKGPU_EINVAL; }
    if (!any) break;
    if (pr[0]->at_ipa()) {
      // the inner-product argument of the whole batch: vectors and folds on the device (ipa_kernels.hpp), one
      // multiscalar-multiplication call per round on the tables, the three transcript messages per round on the host
      const double t0 = now();
      TRY(ipa_on_device(c, ps, pr, parallel));
      if (timing) fprintf(stderr, "prover inner-product argument on the device: %.1f ms\n", (now() - t0) * 1e3);
      continue;
    }
    offs.assign(1, 0);
    row_of[0] = 0;
    for (size_t i = 0; i < batch; ++i) {
      for (const MsmRow& r : rows[i]) offs.push_back(offs.back() + r.scalars.size());
      row_of[i + 1] = offs.size() - 1;
    }
    const double t0 = now();
    sc.resize(32 * offs.back());
    idx.resize(offs.back());
    parallel([&](size_t i) {
      uint64_t k = offs[row_of[i]];
      for (const MsmRow& r : rows[i])
        for (size_t j = 0; j < r.scalars.size(); ++j, ++k) { r.scalars[j].to_bytes(&sc[32 * k]); idx[k] = r.index[j]; }
    });
    const size_t n_rows = offs.size() - 1;
    pts.resize(32 * n_rows);
    const double t1 = now();
    TRY(zkgpu_msm_ps_batch(c, ps, n_rows, sc.data(), idx.data(), offs.data(), pts.data()));
    const double t2 = now();
    parallel([&](size_t i) { if (!pr[i]->done()) pr[i]->step(&pts[32 * row_of[i]], rows[i]); });
    const double t3 = now();
    t_flat += t1 - t0; t_gpu += t2 - t1; t_step += t3 - t2;
    if (timing) fprintf(stderr, "prover phase %d: %zu rows, %llu terms: flatten %.1f ms, msm %.1f ms, host step %.1f ms\n", phase,
                        n_rows, (unsigned long long)offs.back(), (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
  }
  if (timing) fprintf(stderr, "prover total: flatten %.1f ms, msm %.1f ms, host steps %.1f ms (%zu proofs, %d threads)\n",
                      t_flat * 1e3, t_gpu * 1e3, t_step * 1e3, batch, nt);
  const size_t plen = pr[0]->proof().size();
  if (plen > proof_stride) return ZKGPU_EINVAL;
  for (size_t i = 0; i < batch; ++i) {
    if (pr[i]->proof().size() != plen || pr[i]->commitments().size() != com_bytes) return ZKGPU_EINVAL;
    memcpy(commitments + com_bytes * i, pr[i]->commitments().data(), com_bytes);
    memcpy(proofs + proof_stride * i, pr[i]->proof().data(), plen);
  }
  *proof_len = plen;
  return ZKGPU_OK;
}
// ---- slices: one call, several sub-batches in flight --------------------------------------------------------------------
// A proof is a serial chain of ~75 launches in which the multiscalar multiplications on the tables (chip-filling) alternate
// with latency-bound phases -- one workgroup per proof, a Keccak chain or a scalar inversion per proof: 10 - 20 % of the
// vector ALU's issue rate (profiles/pmc_valu_prover.json against the kernel times).  ONE batch on ONE stream therefore leaves
// most of the chip idle most of the time.  A call is cut into contiguous SLICES, each proved on a context of its own (one
// stream, its own workspace; the tables are shared) by a thread of its own: one slice's phases run in the gaps of another's
// multiplication, and the host share of one slice (blinding factors, witness queues, the upload) beside the device share of
// another -- what a caller used to have to arrange with two threads and two contexts (bench.py: two_calls_in_flight).
// Proofs do not depend on the slicing: every proof is a function of its own inputs and seed (tests compare with the oracle).
int prover_slice_count(const zkgpu_ctx* c, size_t batch) {
  static const int env = [] { const char* e = getenv("ZKGPU_PROVER_SLICES"); return e ? std::max(1, std::min(8, atoi(e))) : 0; }();
  const int forced = c->prover_slices ? c->prover_slices : env;
  if (forced) return (int)std::min<size_t>((size_t)forced, std::max<size_t>(1, batch));
  // the sweeps (profiles/archive/r05b_prover_sweep_*.jsonl, r05z_prover_slices_after_stages.jsonl, r05z_prover_bigger_calls.jsonl; DESIGN.md
  // sec 4.4): slices of one to two thousand statements, at most eight
  return batch >= 16384 ? 8 : batch >= 4096 ? 4 : batch >= 2048 ? 3 : batch >= 1024 ? 2 : 1;
}

// one(ctx, lo, hi, host_threads) proves statements [lo, hi) on ctx; c->mu is held by the caller
int run_sliced(zkgpu_ctx* c, size_t batch, int host_threads, const std::function<int(zkgpu_ctx*, size_t, size_t, int)>& one) {
  int S = prover_slice_count(c, batch);
  while ((int)c->pv_slices.size() < S - 1) {
    zkgpu_ctx* t = nullptr;
    DeviceGuard g(c->device);
    if (ctx_create(c->device, nullptr, &t, 2) != ZKGPU_OK) { S = (int)c->pv_slices.size() + 1; break; }   // (fewer slices: still correct)
    c->pv_slices.push_back(t);
  }
  c->last_prover_slices = std::max(S, 1);
  if (S <= 1) return one(c, 0, batch, host_threads);
  const int ht = std::max(1, (host_threads > 0 ? host_threads : usable_cpus()) / S);
  std::vector<int> rc((size_t)S, ZKGPU_OK);
  std::vector<std::thread> th;
  auto cut = [&](int i) { return batch * (size_t)i / (size_t)S; };
  for (int i = 1; i < S; ++i) {
    zkgpu_ctx* t = c->pv_slices[(size_t)i - 1];
    t->prover_mode = c->prover_mode;
    t->profiling = c->profiling;
    auto body = [&, i, t] {
      try { rc[(size_t)i] = one(t, cut(i), cut(i + 1), ht); }
      catch (const std::bad_alloc&) { t->last_error = "prover: out of host memory"; rc[(size_t)i] = ZKGPU_ENOMEM; }
      catch (const std::exception& e) { t->last_error = e.what(); rc[(size_t)i] = ZKGPU_EINVAL; }      // (never across a thread's top frame)
    };
    // a thread that cannot be started (std::system_error: the process is out of threads) must not unwind past the ones
    // already running on the caller's buffers -- that would be std::terminate: its slice runs here instead
    try { th.emplace_back(body); }
    catch (const std::exception&) { body(); }
  }
  try { rc[0] = one(c, 0, cut(1), ht); }
  catch (const std::bad_alloc&) { c->last_error = "prover: out of host memory"; rc[0] = ZKGPU_ENOMEM; }
  catch (const std::exception& e) { c->last_error = e.what(); rc[0] = ZKGPU_EINVAL; }
  for (auto& t : th) t.join();
  int bad = ZKGPU_OK;
  for (int i = 1; i < S; ++i) {
    zkgpu_ctx* t = c->pv_slices[(size_t)i - 1];
    for (const ProfEntry& e : t->prof) { ProfEntry& mine = c->prof[(size_t)prof_index(c, e.name)]; mine.launches += e.launches; mine.ms += e.ms; }
    t->prof.clear();
    if (rc[(size_t)i] != ZKGPU_OK && bad == ZKGPU_OK) { bad = rc[(size_t)i]; c->last_error = t->last_error; }
  }
  return rc[0] != ZKGPU_OK ? rc[0] : bad;
}

int cloak_prove_whole(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t gens_capacity, size_t batch, uint32_t n_in,
                      uint32_t n_out, const uint64_t* quantities, const uint8_t* flavors, const uint8_t* seeds,
                      int host_threads, uint8_t* commitments, uint8_t* proofs, size_t proof_stride,
                      size_t* proof_len) {
  const size_t nv = (size_t)n_in + n_out;
  if (c->prover_mode == 1)
    return prove_lockstep(c, ps, batch, host_threads, [&](size_t i) {
      return cloak_prover(n_in, n_out, quantities + nv * i, flavors + 32 * nv * i, seeds + 32 * i, gens_capacity);
    }, commitments, 64 * nv, proofs, proof_stride, proof_len);
  // the cloak as a described system (traced once), its witness queue per statement, blindings from the seeds
  R1csDesc desc;
  std::vector<uint32_t> md;
  PvHostPlan hp;
  const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  try {
    PvCloakTrace::trace(n_in, n_out, desc, md);
    hp = pv_build(desc, md, gens_capacity);
  } catch (const std::exception& e) { c->last_error = e.what(); return ZKGPU_EINVAL; }
  const double t1 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  const size_t m = 2 * nv;
  std::vector<uint8_t> vals(batch * m * 32), bl(batch * m * 32), giv(batch * (size_t)hp.sh.n_given * 64), rs(batch * 32);
  std::atomic<bool> bad{false};
  host_parallel(batch, host_threads, [&](size_t i) {
    const uint8_t* seed = seeds + 32 * i;
    for (size_t j = 0; j < nv; ++j) {
      Scalar::from_u64(quantities[nv * i + j]).to_bytes(&vals[(i * m + 2 * j) * 32]);
      memcpy(&vals[(i * m + 2 * j + 1) * 32], flavors + 32 * (nv * i + j), 32);
      R1csProver::derive_scalar(seed, "q_blinding", j).to_bytes(&bl[(i * m + 2 * j) * 32]);
      R1csProver::derive_scalar(seed, "f_blinding", j).to_bytes(&bl[(i * m + 2 * j + 1) * 32]);
    }
    std::vector<uint32_t> gw;
    pv_cloak_given(n_in, n_out, quantities + nv * i, flavors + 32 * nv * i, gw);
    if (gw.size() != 16 * (size_t)hp.sh.n_given) { bad = true; return; }
    memcpy(&giv[i * (size_t)hp.sh.n_given * 64], gw.data(), gw.size() * 4);
    R1csProver::derive(seed, "rng", 0, &rs[32 * i], 32);
  });
  if (bad) { c->last_error = "prover: the cloak witness does not fit the traced gadget"; return ZKGPU_EINVAL; }
  if (getenv("ZKGPU_PROVER_TIMING"))
    fprintf(stderr, "device prover: tracing the gadget + tables %.1f ms, blinding factors + witness queues %.1f ms\n", (t1 - t0) * 1e3,
            (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t1) * 1e3);
  return prove_device(c, ps, hp, batch, vals.data(), bl.data(), giv.data(), rs.data(), commitments, proofs, proof_stride, proof_len);
}

// Proves `batch` statements of ONE described constraint system (BASELINE.json configs[4]: "R1CS proving for a
// 1024-constraint program"): the description of zkgpu_r1cs_plan_create plus the witness -- see include/zkgpu.h.
int r1cs_prove_whole(zkgpu_ctx* c, const zkgpu_pointset* ps, const zkgpu_r1cs_desc* d, const uint32_t* mult_def,
                     size_t gens_capacity, size_t batch, const uint8_t* values, const uint8_t* blindings,
                     const uint8_t* given, size_t n_given, const uint8_t* seeds, int host_threads,
                     uint8_t* commitments, uint8_t* proofs, size_t proof_stride, size_t* proof_len) {
  R1csDesc desc;
  if (!desc_from_c(c, d, desc)) return ZKGPU_EINVAL;
  try { (void)plan_from_desc(desc); } catch (const std::exception& e) { c->last_error = e.what(); return ZKGPU_EINVAL; }
  std::vector<uint32_t> md(2 * (size_t)desc.n, MULT_GIVEN);
  if (mult_def) md.assign(mult_def, mult_def + 2 * (size_t)desc.n);
  const size_t m = desc.m;
  if (c->prover_mode == 1)
    return prove_lockstep(c, ps, batch, host_threads, [&](size_t i) {
      std::vector<Scalar> vals, bl;
      for (size_t j = 0; j < m; ++j) {
        uint8_t wide[64] = {0};
        memcpy(wide, values + 32 * (m * i + j), 32);
        vals.push_back(Scalar::from_wide(wide));
        if (blindings) { uint8_t wb[64] = {0}; memcpy(wb, blindings + 32 * (m * i + j), 32); bl.push_back(Scalar::from_wide(wb)); }
        else bl.push_back(R1csProver::derive_scalar(seeds + 32 * i, "blinding", j));
      }
      std::vector<std::pair<Scalar, Scalar>> gv;
      for (size_t j = 0; j < n_given; ++j) {
        uint8_t wl[64] = {0}, wr[64] = {0};
        memcpy(wl, given + 64 * (n_given * i + j), 32); memcpy(wr, given + 64 * (n_given * i + j) + 32, 32);
        gv.emplace_back(Scalar::from_wide(wl), Scalar::from_wide(wr));
      }
      return desc_prover(desc, md, std::move(vals), std::move(bl), gv, seeds + 32 * i, gens_capacity);
    }, commitments, 32 * m, proofs, proof_stride, proof_len);
  PvHostPlan hp;
  try { hp = pv_build(desc, md, gens_capacity); } catch (const std::exception& e) { c->last_error = e.what(); return ZKGPU_EINVAL; }
  if (hp.sh.n_given != n_given) { c->last_error = "prover: n_given does not match the multipliers without defining constraints"; return ZKGPU_EINVAL; }
  std::vector<uint8_t> bl, rs(batch * 32);
  if (!blindings) bl.resize(batch * m * 32);
  host_parallel(batch, host_threads, [&](size_t i) {
    if (!blindings) for (size_t j = 0; j < m; ++j) R1csProver::derive_scalar(seeds + 32 * i, "blinding", j).to_bytes(&bl[(i * m + j) * 32]);
    R1csProver::derive(seeds + 32 * i, "rng", 0, &rs[32 * i], 32);
  });
  return prove_device(c, ps, hp, batch, values, blindings ? blindings : bl.data(), given, rs.data(), commitments, proofs, proof_stride, proof_len);
}
}  // namespace

int zkgpu_cloak_prove_batch(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t gens_capacity, size_t batch, uint32_t n_in,
                            uint32_t n_out, const uint64_t* quantities, const uint8_t* flavors, const uint8_t* seeds,
                            int host_threads, uint8_t* commitments, uint8_t* proofs, size_t proof_stride,
                            size_t* proof_len) {
  if (!c || !ps || !commitments || !proofs || !proof_len) return ZKGPU_EINVAL;
  *proof_len = 0;
  if (batch == 0) return ZKGPU_OK;
  if (!quantities || !flavors || !seeds || ps->n < 2 + 2 * gens_capacity || n_in + n_out == 0) return ZKGPU_EINVAL;
  if (!ps->table) { c->last_error = "zkgpu_cloak_prove_batch needs zkgpu_pointset_build_tables first"; return ZKGPU_EINVAL; }
  const size_t nv = (size_t)n_in + n_out;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (c->prover_mode == 1)
    return cloak_prove_whole(c, ps, gens_capacity, batch, n_in, n_out, quantities, flavors, seeds, host_threads, commitments, proofs, proof_stride, proof_len);
  std::atomic<size_t> len_out{0};
  const int rc = run_sliced(c, batch, host_threads, [&](zkgpu_ctx* t, size_t lo, size_t hi, int ht) {
    size_t pl = 0;
    const int r = cloak_prove_whole(t, ps, gens_capacity, hi - lo, n_in, n_out, quantities + nv * lo, flavors + 32 * nv * lo, seeds + 32 * lo, ht,
                                    commitments + 64 * nv * lo, proofs + proof_stride * lo, proof_stride, &pl);
    if (r == ZKGPU_OK) len_out = pl;
    return r;
  });
  if (rc != ZKGPU_OK) { memset(commitments, 0, 64 * nv * batch); memset(proofs, 0, proof_stride * batch); return rc; }
  *proof_len = len_out;
  return ZKGPU_OK;
}

int zkgpu_r1cs_prove_batch(zkgpu_ctx* c, const zkgpu_pointset* ps, const zkgpu_r1cs_desc* d, const uint32_t* mult_def,
                           size_t gens_capacity, size_t batch, const uint8_t* values, const uint8_t* blindings,
                           const uint8_t* given, size_t n_given, const uint8_t* seeds, int host_threads,
                           uint8_t* commitments, uint8_t* proofs, size_t proof_stride, size_t* proof_len) {
  if (!c || !ps || !d || !commitments || !proofs || !proof_len) return ZKGPU_EINVAL;
  *proof_len = 0;
  if (batch == 0) return ZKGPU_OK;
  if (!seeds || ps->n < 2 + 2 * gens_capacity || (d->n_commitments && !values) || (n_given && !given)) return ZKGPU_EINVAL;
  if (!ps->table) { c->last_error = "zkgpu_r1cs_prove_batch needs zkgpu_pointset_build_tables first"; return ZKGPU_EINVAL; }
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (c->prover_mode == 1)
    return r1cs_prove_whole(c, ps, d, mult_def, gens_capacity, batch, values, blindings, given, n_given, seeds, host_threads, commitments, proofs, proof_stride, proof_len);
  const size_t m = d->n_commitments;
  std::atomic<size_t> len_out{0};
  const int rc = run_sliced(c, batch, host_threads, [&](zkgpu_ctx* t, size_t lo, size_t hi, int ht) {
    size_t pl = 0;
    const int r = r1cs_prove_whole(t, ps, d, mult_def, gens_capacity, hi - lo, values ? values + 32 * m * lo : nullptr,
                                   blindings ? blindings + 32 * m * lo : nullptr, given ? given + 64 * n_given * lo : nullptr, n_given,
                                   seeds + 32 * lo, ht, commitments + 32 * m * lo, proofs + proof_stride * lo, proof_stride, &pl);
    if (r == ZKGPU_OK) len_out = pl;
    return r;
  });
  if (rc != ZKGPU_OK) { memset(commitments, 0, 32 * m * batch); memset(proofs, 0, proof_stride * batch); return rc; }
  *proof_len = len_out;
  return ZKGPU_OK;
}

int zkgpu_decode_check(zkgpu_ctx* c, const uint8_t* points, size_t n, uint8_t* ok) {
  if (!c || (n && (!points || !ok))) return ZKGPU_EINVAL;
  if (n == 0) return ZKGPU_OK;
  memset(ok, 0, n);
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  TRY(upload(c, c->in_points, points, n * 32));
  TRY(ensure(c, c->ok_bytes, n));
  {
    Launch l(c, "k_decompress");
    hipLaunchKernelGGL(k_decompress, dim3(blocks_for(n, 256)), dim3(256), 0, c->stream,
                       (const uint32_t*)c->in_points.p, (uint32_t*)nullptr, (uint64_t)n, (const uint64_t*)nullptr,
                       1u, (uint32_t*)nullptr, (unsigned long long*)nullptr, (uint8_t*)c->ok_bytes.p);
  }
  HIP_TRY(c, hipMemcpyAsync(ok, c->ok_bytes.p, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->profiling) prof_collect(c);
  return ZKGPU_OK;
}

// ---- proof bytes in, accept bits out -------------------------------------------------
// Host side of r1cs::Verifier::verify for ZkVM `cloak` statements (r1cs_verifier.hpp),
// spread over `host_threads` threads, then ONE device call for the whole batch.
namespace {
struct CloakBatch {
  std::vector<uint64_t> dyn_off, st_off;
  std::vector<uint8_t> dyn_sc, dyn_pt, st_sc, wellformed;
  std::vector<uint32_t> st_idx;
};

int prepare_cloak_batch(size_t gens_capacity, size_t batch, const uint32_t* n_in, const uint32_t* n_out,
                        const uint8_t* commitments, const uint8_t* proofs, const uint64_t* proof_offsets,
                        const uint8_t* r_bytes, int host_threads, CloakBatch& out) {
  std::vector<uint64_t> com_off(batch + 1, 0);
  for (size_t i = 0; i < batch; ++i) com_off[i + 1] = com_off[i] + 64ull * ((uint64_t)n_in[i] + n_out[i]);
  std::vector<VerifierMsm> prep(batch);
  out.wellformed.assign(batch, 0);
  // verifier randomness r: caller-provided (reproducible) or from the OS
  std::vector<uint8_t> rnd;
  if (!r_bytes) {
    rnd.resize(64 * batch);
    if (!os_random(rnd.data(), rnd.size())) return ZKGPU_EINVAL;
    r_bytes = rnd.data();
  }
  const int nt = std::max(1, std::min<int>(host_threads > 0 ? host_threads : usable_cpus(), 256));
  auto work = [&](int tid) {
    for (size_t i = (size_t)tid; i < batch; i += (size_t)nt) {
      if (proof_offsets[i + 1] < proof_offsets[i]) continue;
      const Scalar r = Scalar::from_wide(r_bytes + 64 * i);
      out.wellformed[i] = cloak::prepare_tx(commitments + com_off[i], n_in[i], n_out[i], proofs + proof_offsets[i],
                                            (size_t)(proof_offsets[i + 1] - proof_offsets[i]), r, gens_capacity,
                                            prep[i]) ? 1 : 0;
    }
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& t : th) t.join();
  }
  // CSR over the well-formed proofs (malformed ones keep an empty row and are masked out by the caller)
  out.dyn_off.assign(batch + 1, 0);
  out.st_off.assign(batch + 1, 0);
  for (size_t i = 0; i < batch; ++i) {
    out.dyn_off[i + 1] = out.dyn_off[i] + (out.wellformed[i] ? prep[i].dyn_scalars.size() / 32 : 0);
    out.st_off[i + 1] = out.st_off[i] + (out.wellformed[i] ? prep[i].static_scalars.size() / 32 : 0);
  }
  out.dyn_sc.resize(32 * out.dyn_off[batch]);
  out.dyn_pt.resize(32 * out.dyn_off[batch]);
  out.st_sc.resize(32 * out.st_off[batch]);
  out.st_idx.resize(out.st_off[batch]);
  for (size_t i = 0; i < batch; ++i) {
    if (!out.wellformed[i]) continue;
    memcpy(&out.dyn_sc[32 * out.dyn_off[i]], prep[i].dyn_scalars.data(), prep[i].dyn_scalars.size());
    memcpy(&out.dyn_pt[32 * out.dyn_off[i]], prep[i].dyn_points.data(), prep[i].dyn_points.size());
    memcpy(&out.st_sc[32 * out.st_off[i]], prep[i].static_scalars.data(), prep[i].static_scalars.size());
    memcpy(&out.st_idx[out.st_off[i]], prep[i].static_index.data(), prep[i].static_index.size() * 4);
  }
  return ZKGPU_OK;
}
}  // namespace

int zkgpu_cloak_prepare_batch(size_t gens_capacity, size_t batch, const uint32_t* n_in, const uint32_t* n_out,
                              const uint8_t* commitments, const uint8_t* proofs, const uint64_t* proof_offsets,
                              const uint8_t* r_bytes, int host_threads, uint8_t* dyn_scalars, uint8_t* dyn_points,
                              uint64_t* dyn_offsets, size_t dyn_capacity, uint8_t* static_scalars,
                              uint32_t* static_index, uint64_t* static_offsets, size_t static_capacity,
                              uint8_t* wellformed) {
  if (batch && (!n_in || !n_out || !commitments || !proofs || !proof_offsets)) return ZKGPU_EINVAL;
  if (!dyn_offsets || !static_offsets || !wellformed) return ZKGPU_EINVAL;
  CloakBatch cb;
  TRY(prepare_cloak_batch(gens_capacity, batch, n_in, n_out, commitments, proofs, proof_offsets, r_bytes,
                          host_threads, cb));
  if (cb.dyn_off[batch] > dyn_capacity || cb.st_off[batch] > static_capacity) return ZKGPU_EINVAL;
  memcpy(dyn_offsets, cb.dyn_off.data(), (batch + 1) * 8);
  memcpy(static_offsets, cb.st_off.data(), (batch + 1) * 8);
  memcpy(wellformed, cb.wellformed.data(), batch);
  if (!cb.dyn_sc.empty()) { memcpy(dyn_scalars, cb.dyn_sc.data(), cb.dyn_sc.size()); memcpy(dyn_points, cb.dyn_pt.data(), cb.dyn_pt.size()); }
  if (!cb.st_sc.empty()) { memcpy(static_scalars, cb.st_sc.data(), cb.st_sc.size()); memcpy(static_index, cb.st_idx.data(), cb.st_idx.size() * 4); }
  return ZKGPU_OK;
}

// ---- the same, with the host half moved onto the device (plan replay) -------------------
struct zkgpu_cloak_plan {
  zkgpu_ctx* ctx;
  int device = 0;
  std::mutex mu;       // guards the per-batch-size scaffolding below
  CloakPlan host;
  PrepShape shape;
  size_t gens_capacity;
  uint32_t *d_init = nullptr, *d_mono_chal = nullptr, *d_mono_pow = nullptr, *d_tgt_off = nullptr, *d_term_q = nullptr,
           *d_term_mono = nullptr, *d_term_coef = nullptr;
  uint32_t* d_tape = nullptr;     // transcript_tape.hpp, four words per operation
  uint32_t n_ops = 0;
  uint32_t *d_seg_info = nullptr, *d_seg_const = nullptr;   // the tape regrouped for k_transcript_coop
  uint16_t* d_seg_map = nullptr;
  uint32_t n_seg = 0;
  // CSR scaffolding of a uniform batch (offsets, generator index template) for the largest batch seen so
  // far; a prefix of it serves every smaller batch.  It only grows: a larger one is built beside the old
  // one, which batches in flight on other contexts may still be reading and which is therefore kept
  // until the plan is destroyed (`retired`).  Read the three pointers under `mu`.
  size_t cached_batch = 0;
  uint64_t *d_dyn_off = nullptr, *d_st_off = nullptr;
  uint32_t* d_st_index = nullptr;
  std::vector<void*> retired;
  size_t lds_bytes = 0;
};

namespace {
int plan_upload_bytes(zkgpu_ctx* c, void** dst, const void* src, size_t bytes) {
  {
    const hipError_t e = hipMalloc(dst, std::max<size_t>(bytes, 16));
    if (e != hipSuccess) { *dst = nullptr; c->last_error = std::string("hipMalloc (plan): ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? ZKGPU_ENOMEM : ZKGPU_EHIP; }
  }
  if (bytes) HIP_TRY(c, hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
  return ZKGPU_OK;
}
#define plan_upload(c, dst, vec) plan_upload_bytes((c), (void**)(dst), (vec).data(), (vec).size() * sizeof((vec)[0]))
}  // namespace

namespace {
bool desc_from_c(zkgpu_ctx* c, const zkgpu_r1cs_desc* d, R1csDesc& desc);
// common tail of plan creation: p->host is set.  Error codes matter to the callers that cache plans per shape
// (session.hpp, verifier_plan): ZKGPU_EINVAL = this statement can never be verified over this generator set (what the
// reference answers with InvalidGeneratorsLength); anything else (ZKGPU_ENOMEM, ZKGPU_EHIP) is transient.
int plan_finish_inner(zkgpu_ctx* c, zkgpu_cloak_plan* p, size_t gens_capacity) {
  const CloakPlan& h = p->host;
  if (h.pn > gens_capacity || h.k > 16) { c->last_error = "statement needs more generators than the set holds"; return ZKGPU_EINVAL; }
  PrepShape& s = p->shape;
  s.m = h.m; s.n1 = h.n1; s.n = h.n; s.pn = h.pn; s.k = h.k; s.n_cons = h.n_cons;
  s.n_chal2 = (uint32_t)h.chal_names.size();
  s.n_mono = (uint32_t)h.mono_chal.size();
  s.n_targets = h.n_targets();
  s.n_terms = (uint32_t)h.term_q.size();
  s.proof_words = (16 + 2 * h.k) * 8;
  s.n_ch = CH_FIXED + s.n_chal2 + 2 * h.k;
  s.n_dyn = 11 + h.m + 2 * h.k;
  s.n_static = 2 + 2 * h.pn;
  s.n_heavy = 0;
  for (uint32_t t = 0; t < s.n_targets && s.n_heavy < 8; ++t)
    if (h.tgt_off[t + 1] - h.tgt_off[t] > 2 * HEAVY_TERMS) s.heavy[s.n_heavy++] = t;
  s.n_ch_ext = s.n_ch + s.n_mono + PREP_STRIDES;
  // the fewest flattening passes (<= 5) that let as many workgroups share a CU's 160 KB of LDS as any number of passes
  // would (a payment: 45.9 KB in one pass, 42.3 KB in three or more -- three workgroups either way: one pass); the split
  // points are target boundaries.  (Every pass costs the workgroup two barriers and a round of dependent loads.)
  const uint32_t n_prod = (uint32_t)h.prod_q.size();
  PrepShape best = s;
  size_t best_bytes = 0, best_groups = 0;
  for (uint32_t chunks = 1; chunks <= 5; ++chunks) {
    s.n_chunks = chunks;
    s.chunk_tgt[0] = 0;
    s.chunk_prod[0] = 0;
    uint32_t cap = 0, g = 0;
    for (uint32_t ck = 0; ck < chunks; ++ck) {
      const uint32_t goal = (uint32_t)(((uint64_t)n_prod * (ck + 1)) / chunks);
      while (g < s.n_targets && (h.prod_off[g + 1] <= goal || ck + 1 == chunks)) ++g;
      if (ck + 1 == chunks) g = s.n_targets;
      s.chunk_tgt[ck + 1] = g;
      s.chunk_prod[ck + 1] = h.prod_off[g];
      cap = std::max(cap, h.prod_off[g] - h.prod_off[s.chunk_tgt[ck]]);
    }
    s.tv_cap = cap;
    const size_t bytes = prepare_lds_bytes(s), groups = std::min<size_t>(4, (160 * 1024) / bytes);     // (four: the kernel's launch bounds)
    if (best_bytes == 0 || groups > best_groups || (groups == 0 && bytes < best_bytes)) { best = s; best_bytes = bytes; best_groups = groups; }
    if (groups == 4) break;
  }
  s = best;
  p->lds_bytes = best_bytes;
  if (p->lds_bytes > 160 * 1024) { c->last_error = "plan does not fit the 160 KiB LDS of a CU"; return ZKGPU_EINVAL; }
  // STROBE state after Transcript::new("ZkVM.r1cs") + r1cs_domain_sep()
  Transcript tr(h.label.c_str());
  tr.append_message("dom-sep", (const uint8_t*)"r1cs v1", 7);
  std::vector<uint32_t> init(52);
  tr.export_state(init.data());
  TRY(plan_upload(c, &p->d_init, init));
  {
    const std::vector<uint32_t> tape = build_r1cs_verifier_tape(init[50], init[51], s.m, h.chal_names, s.k, s.pn, CH_FIXED);
    p->n_ops = (uint32_t)(tape.size() / 4);
    TRY(plan_upload(c, &p->d_tape, tape));
    const CoopSegments segs = build_coop_segments(tape, s.m);
    if (segs.n_seg() && s.n_ch <= 0xffffu) {
      p->n_seg = segs.n_seg();
      TRY(plan_upload(c, &p->d_seg_info, segs.info));
      TRY(plan_upload(c, &p->d_seg_const, segs.consts));
      TRY(plan_upload(c, &p->d_seg_map, segs.map));
    }
  }
  TRY(plan_upload(c, &p->d_mono_chal, h.mono_chal));
  TRY(plan_upload(c, &p->d_mono_pow, h.mono_pow));
  TRY(plan_upload(c, &p->d_tgt_off, h.tgt_off));
  {
    std::vector<uint32_t> qm(2 * h.prod_q.size());
    for (size_t i = 0; i < h.prod_q.size(); ++i) { qm[2 * i] = h.prod_q[i]; qm[2 * i + 1] = h.prod_mono[i]; }
    TRY(plan_upload(c, &p->d_term_q, h.term_info));       // k_prepare's term_info
    TRY(plan_upload(c, &p->d_term_mono, qm));             //             prod_qm
    TRY(plan_upload(c, &p->d_term_coef, h.prod_coef));    //             prod_coef
  }
  HIP_TRY(c, hipFuncSetAttribute((const void*)k_prepare, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
  return ZKGPU_OK;
}

int plan_finish(zkgpu_ctx* c, zkgpu_cloak_plan* p, size_t gens_capacity, zkgpu_cloak_plan** out) {
  const int rc = plan_finish_inner(c, p, gens_capacity);
  if (rc != ZKGPU_OK) { zkgpu_cloak_plan_destroy(p); return rc; }     // whatever was uploaded so far goes with it
  *out = p;
  return ZKGPU_OK;
}

}  // namespace

int zkgpu_cloak_plan_create(zkgpu_ctx* c, uint32_t n_in, uint32_t n_out, size_t gens_capacity, zkgpu_cloak_plan** out) {
  if (!c || !out || n_in + n_out == 0 || n_in > 64 || n_out > 64) return ZKGPU_EINVAL;
  *out = nullptr;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  zkgpu_cloak_plan* p = new zkgpu_cloak_plan();
  p->ctx = c;
  p->device = c->device;
  p->gens_capacity = gens_capacity;
  try {
    p->host = PlanBuilder::build(n_in, n_out);     // the cloak gadget traced into a description, then the generic path
  } catch (const std::bad_alloc&) {                // out of host memory THIS time: not "the shape can never be verified"
    c->last_error = "out of host memory while tracing the cloak gadget";
    delete p;
    return ZKGPU_ENOMEM;
  } catch (const std::exception& e) {
    c->last_error = e.what();
    delete p;
    return ZKGPU_EINVAL;
  }
  return plan_finish(c, p, gens_capacity, out);
}

// A plan for ANY constraint system, described as data (what lets Tx::verify use the
// device path for statements that are not a pure cloak).  See include/zkgpu.h for the conventions.
int zkgpu_r1cs_plan_create(zkgpu_ctx* c, const zkgpu_r1cs_desc* d, size_t gens_capacity, zkgpu_cloak_plan** out) {
  if (!c || !out || !d) return ZKGPU_EINVAL;
  *out = nullptr;
  R1csDesc desc;
  if (!desc_from_c(c, d, desc)) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  zkgpu_cloak_plan* p = new zkgpu_cloak_plan();
  p->ctx = c;
  p->device = c->device;
  p->gens_capacity = gens_capacity;
  try {
    p->host = plan_from_desc(desc);
  } catch (const std::exception& e) {
    c->last_error = e.what();
    delete p;
    return ZKGPU_EINVAL;
  }
  return plan_finish(c, p, gens_capacity, out);
}

// the whole-proof entry points under their generic names: a plan is a plan, whatever statement it was made from;
// commitments = m x 32 bytes per statement
int zkgpu_r1cs_verify_batch_gpu(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch, const uint8_t* commitments,
                                const uint8_t* proofs, size_t proof_len, const uint8_t* r_bytes, uint8_t* accept_bitmap) {
  return zkgpu_cloak_verify_batch_gpu(c, ps, plan, batch, commitments, proofs, proof_len, r_bytes, accept_bitmap);
}
int zkgpu_r1cs_verify_submit(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch, const uint8_t* commitments,
                             const uint8_t* proofs, size_t proof_len, const uint8_t* r_bytes) {
  return zkgpu_cloak_verify_submit(c, ps, plan, batch, commitments, proofs, proof_len, r_bytes);
}
int zkgpu_r1cs_verify_submit_dev(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch, const void* d_commitments,
                                 const void* d_proofs, size_t proof_len, const void* d_r) {
  return zkgpu_cloak_verify_submit_dev(c, ps, plan, batch, d_commitments, d_proofs, proof_len, d_r);
}
void zkgpu_r1cs_plan_destroy(zkgpu_cloak_plan* p) { zkgpu_cloak_plan_destroy(p); }

void zkgpu_cloak_plan_destroy(zkgpu_cloak_plan* p) {
  if (!p) return;
  DeviceGuard g(p->ctx->device);
  void* ptrs[] = {p->d_seg_info, p->d_seg_const, p->d_seg_map, p->d_tape, p->d_init, p->d_mono_chal, p->d_mono_pow, p->d_tgt_off, p->d_term_q, p->d_term_mono,
                  p->d_term_coef, p->d_dyn_off, p->d_st_off, p->d_st_index};
  for (void* q : ptrs) if (q) (void)hipFree(q);
  for (void* q : p->retired) (void)hipFree(q);
  delete p;
}

int zkgpu_cloak_plan_info(const zkgpu_cloak_plan* p, uint32_t* multipliers, uint32_t* padded_n, uint32_t* constraints,
                          uint32_t* terms, uint32_t* proof_len) {
  if (!p) return ZKGPU_EINVAL;
  if (multipliers) *multipliers = p->host.n;
  if (padded_n) *padded_n = p->host.pn;
  if (constraints) *constraints = p->host.n_cons;
  if (terms) *terms = (uint32_t)p->host.term_q.size();
  if (proof_len) *proof_len = 1 + 32 * (16 + 2 * p->host.k);
  return ZKGPU_OK;
}

// Buffer layout of the device-side preparation for this plan (what zkgpu_debug_read returns):
// out[0] slots per transaction in "challenges", [1] challenge slots proper, [2] second-phase challenges,
// [3] dynamic terms, [4] static terms, [5] k, [6] commitments m, [7] monomials.
int zkgpu_cloak_plan_layout(const zkgpu_cloak_plan* p, uint32_t out[8]) {
  if (!p || !out) return ZKGPU_EINVAL;
  const PrepShape& s = p->shape;
  out[0] = s.n_ch_ext; out[1] = s.n_ch; out[2] = s.n_chal2; out[3] = s.n_dyn; out[4] = s.n_static; out[5] = s.k;
  out[6] = s.m; out[7] = s.n_mono;
  return ZKGPU_OK;
}

// Proof bytes in, accept bits out, everything after the PCIe copy on the device: transcript
// replay (one lane per transaction), scalar preparation (one workgroup per transaction),
// then the multiscalar multiplications.  All `batch` statements have the plan's shape.
namespace {
// shared body: d_com / d_proofs / d_r are device pointers
int cloak_verify_gpu_enqueue(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch,
                             const uint32_t* d_com, const uint8_t* d_proofs, const uint32_t* d_r, size_t proof_len);
int cloak_verify_gpu_body(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch,
                          const uint32_t* d_com, const uint8_t* d_proofs, const uint32_t* d_r, size_t proof_len,
                          uint8_t* accept_bitmap);
}  // namespace

namespace {
// host buffers -> pinned staging -> device, queued on the context's light stream (no host wait: the
// caller's buffers are free again when this returns, the copies overlap the batches in flight)
int stage_inputs(zkgpu_ctx* c, const PrepShape& sh, size_t batch, const uint8_t* commitments, const uint8_t* proofs,
                 size_t proof_len, const uint8_t* r_bytes) {
  const size_t n_com = batch * sh.m * 32, n_pr = batch * proof_len, n_r = batch * 64;
  const size_t o_pr = (n_com + 255) & ~(size_t)255, o_r = (o_pr + n_pr + 255) & ~(size_t)255;
  if (c->pinned_in_cap < o_r + n_r) {
    if (c->pinned_in) HIP_TRY(c, hipHostFree(c->pinned_in));
    c->pinned_in = nullptr; c->pinned_in_cap = 0;
    HIP_TRY(c, hipHostMalloc(&c->pinned_in, o_r + n_r + 4096, hipHostMallocDefault));
    c->pinned_in_cap = o_r + n_r + 4096;
  }
  char* h = (char*)c->pinned_in;
  memcpy(h, commitments, n_com);
  memcpy(h + o_pr, proofs, n_pr);
  if (r_bytes) {
    memcpy(h + o_r, r_bytes, n_r);
  } else if (!os_random(h + o_r, n_r)) {   // verifier randomness from the OS
    c->last_error = "getrandom failed";
    return ZKGPU_EINVAL;
  }
  TRY(ensure(c, c->prep_com, std::max<size_t>(n_com, 16)));
  TRY(ensure(c, c->prep_proofs, std::max<size_t>(n_pr, 16)));
  TRY(ensure(c, c->prep_r, std::max<size_t>(n_r, 16)));
  HIP_TRY(c, hipMemcpyAsync(c->prep_com.p, h, n_com, hipMemcpyHostToDevice, c->stream_l));
  HIP_TRY(c, hipMemcpyAsync(c->prep_proofs.p, h + o_pr, n_pr, hipMemcpyHostToDevice, c->stream_l));
  HIP_TRY(c, hipMemcpyAsync(c->prep_r.p, h + o_r, n_r, hipMemcpyHostToDevice, c->stream_l));
  return ZKGPU_OK;
}
}  // namespace

// Asynchronous form of zkgpu_cloak_verify_batch_gpu (inputs in host memory; they may be reused as
// soon as this returns); zkgpu_verify_wait collects the bitmap.
int zkgpu_cloak_verify_submit(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch,
                              const uint8_t* commitments, const uint8_t* proofs, size_t proof_len,
                              const uint8_t* r_bytes) {
  if (!c || !ps || !plan || plan->device != c->device || ps->ctx->device != c->device) return ZKGPU_EINVAL;
  if (batch == 0 || !commitments || !proofs || batch >= (1ull << 24)) return ZKGPU_EINVAL;
  const PrepShape& sh = plan->shape;
  {
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->pending) return ZKGPU_EINVAL;
    if (!proof_len_fits(sh, proof_len)) {     // wrong length for this statement: every proof is Err
      std::vector<uint8_t> z((batch + 7) / 8, 0);
      park_sync_result(c, ZKGPU_OK, z.data(), batch);
      return ZKGPU_OK;
    }
    DeviceGuard g(c->device);
    TRY(stage_inputs(c, sh, batch, commitments, proofs, proof_len, r_bytes));
  }
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  return cloak_verify_gpu_enqueue(c, ps, plan, batch, (const uint32_t*)c->prep_com.p, (const uint8_t*)c->prep_proofs.p,
                                  (const uint32_t*)c->prep_r.p, proof_len);
}

int zkgpu_cloak_verify_batch_gpu(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch,
                                 const uint8_t* commitments, const uint8_t* proofs, size_t proof_len,
                                 const uint8_t* r_bytes, uint8_t* accept_bitmap) {
  if (!c || !ps || !plan || plan->device != c->device || ps->ctx->device != c->device || !accept_bitmap) return ZKGPU_EINVAL;
  memset(accept_bitmap, 0, (batch + 7) / 8);
  if (batch == 0) return ZKGPU_OK;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(zkgpu_cloak_verify_submit(c, ps, plan, batch, commitments, proofs, proof_len, r_bytes));
  DeviceGuard g(c->device);
  int rc = pipe_wait(c, accept_bitmap);
  if (rc != ZKGPU_OK) memset(accept_bitmap, 0, (batch + 7) / 8);
  return rc;
}

// Inputs already resident in HBM (what bench.py times): d_commitments = batch x 64 (n_in + n_out)
// bytes, d_proofs = batch x proof_len bytes, d_r = batch x 64 bytes of verifier randomness.
int zkgpu_cloak_verify_batch_gpu_dev(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch,
                                     const void* d_commitments, const void* d_proofs, size_t proof_len,
                                     const void* d_r, uint8_t* accept_bitmap) {
  if (!c || !ps || !plan || plan->device != c->device || ps->ctx->device != c->device || !accept_bitmap) return ZKGPU_EINVAL;
  memset(accept_bitmap, 0, (batch + 7) / 8);
  if (batch == 0) return ZKGPU_OK;
  if (!d_commitments || !d_proofs || !d_r || batch >= (1ull << 24)) return ZKGPU_EINVAL;
  if (!proof_len_fits(plan->shape, proof_len)) return ZKGPU_OK;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  return cloak_verify_gpu_body(c, ps, plan, batch, (const uint32_t*)d_commitments, (const uint8_t*)d_proofs,
                               (const uint32_t*)d_r, proof_len, accept_bitmap);
}

namespace {
// Enqueues (pipeline) or runs (shapes the pipeline does not cover) the verification of one batch;
// the result is collected by pipe_wait.
int cloak_verify_gpu_enqueue(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch,
                             const uint32_t* d_com, const uint8_t* d_proofs, const uint32_t* d_r, size_t proof_len) {
  if (ps->n < 2 + 2 * plan->gens_capacity) return ZKGPU_EINVAL;
  if (c->pending) { c->last_error = "a submitted batch is still waiting for zkgpu_verify_wait"; return ZKGPU_EINVAL; }
  if (c->dep_event) {            // inputs on their way to HBM on another stream (session.hpp, staged transactions)
    const hipEvent_t ev = c->dep_event;
    c->dep_event = nullptr;
    HIP_TRY(c, hipStreamWaitEvent(c->stream_l, ev, 0));
  }
  const PrepShape& sh = plan->shape;
  const uint32_t B = (uint32_t)batch;
  TRY(ensure(c, c->prep_pw, (size_t)B * sh.proof_words * 4));
  TRY(ensure(c, c->prep_ch, (size_t)B * sh.n_ch_ext * 32));
  TRY(ensure(c, c->prep_wf, (size_t)B * 4));
  TRY(ensure(c, c->prep_dyn_sc, (size_t)B * sh.n_dyn * 32));
  TRY(ensure(c, c->prep_dyn_pt, (size_t)B * sh.n_dyn * 32));
  TRY(ensure(c, c->prep_st_sc, (size_t)B * sh.n_static * 32));
  if (plan->n_seg && (c->transcript_mode == 2 || (c->transcript_mode == 0 && batch <= COOP_TRANSCRIPT_MAX))) {
    TRY(ensure(c, c->prep_absorb, (size_t)B * plan->n_seg * 25 * 8));
    TRY(ensure(c, c->prep_raw, (size_t)B * sh.n_ch * 64));
  }
  const uint64_t *d_dyn_off, *d_st_off;
  const uint32_t* d_st_index;
  {
    std::lock_guard<std::mutex> plk(plan->mu);
    if (plan->cached_batch < batch) {
      const size_t nb = std::max(batch, 2 * plan->cached_batch);
      std::vector<uint64_t> doff(nb + 1), soff(nb + 1);
      for (size_t i = 0; i <= nb; ++i) { doff[i] = i * sh.n_dyn; soff[i] = i * sh.n_static; }
      std::vector<uint32_t> idx((size_t)nb * sh.n_static);
      for (size_t i = 0; i < nb; ++i) {
        uint32_t* row = &idx[i * sh.n_static];
        row[0] = 0; row[1] = 1;
        for (uint32_t j = 0; j < sh.pn; ++j) { row[2 + j] = 2 + j; row[2 + sh.pn + j] = (uint32_t)(2 + plan->gens_capacity + j); }
      }
      uint64_t *nd = nullptr, *ns = nullptr;
      uint32_t* ni = nullptr;
      int rc = plan_upload(c, &nd, doff);
      if (rc == ZKGPU_OK) rc = plan_upload(c, &ns, soff);
      if (rc == ZKGPU_OK) rc = plan_upload(c, &ni, idx);
      if (rc != ZKGPU_OK) {
        if (nd) (void)hipFree(nd);
        if (ns) (void)hipFree(ns);
        if (ni) (void)hipFree(ni);
        return rc;
      }
      if (plan->d_dyn_off) { plan->retired.push_back(plan->d_dyn_off); plan->retired.push_back(plan->d_st_off); plan->retired.push_back(plan->d_st_index); }
      plan->d_dyn_off = nd; plan->d_st_off = ns; plan->d_st_index = ni;
      plan->cached_batch = nb;
    }
    d_dyn_off = plan->d_dyn_off; d_st_off = plan->d_st_off; d_st_index = plan->d_st_index;
  }
  Job job;
  job.d_dyn_scalars = (const uint32_t*)c->prep_dyn_sc.p;
  job.d_dyn_points = (const uint32_t*)c->prep_dyn_pt.p;
  job.d_dyn_offsets = d_dyn_off;
  job.n_dyn = (uint64_t)B * sh.n_dyn;
  job.d_st_scalars = (const uint32_t*)c->prep_st_sc.p;
  job.d_st_index = d_st_index;
  job.d_st_offsets = d_st_off;
  job.n_static = (uint64_t)B * sh.n_static;
  job.d_static_rows = ps->rows;
  job.n_msm = B;
  job.d_wellformed = (const uint32_t*)c->prep_wf.p;   // folded into the accept bitmap on the device
  if (pipe_eligible(c, job, ps)) {
    PrepLaunch pl;
    pl.sh = sh;
    pl.d_init = plan->d_init; pl.d_mono_chal = plan->d_mono_chal; pl.d_mono_pow = plan->d_mono_pow;
    pl.d_tgt_off = plan->d_tgt_off; pl.d_term_q = plan->d_term_q; pl.d_term_mono = plan->d_term_mono;
    pl.d_term_coef = plan->d_term_coef; pl.d_tape = plan->d_tape; pl.n_ops = plan->n_ops; pl.lds_bytes = plan->lds_bytes;
    pl.d_seg_info = plan->d_seg_info; pl.d_seg_const = plan->d_seg_const; pl.d_seg_map = plan->d_seg_map; pl.n_seg = plan->n_seg;
    pl.d_com = d_com; pl.d_proofs = d_proofs; pl.d_r = d_r; pl.proof_len = proof_len;
    return pipe_enqueue(c, job, ps, &pl);
  }
  // general shapes (no generator tables, forced window width, many proof points): one stream, synchronous
  hipStream_t s = c->stream;
  TRY(ensure(c, c->recoded, std::max<uint64_t>(job.n_dyn, 1) * 32));
  if (c->reserve_only) return ZKGPU_OK;
  HIP_TRY(c, hipStreamSynchronize(c->stream_l));      // uploads, if any, were queued on the light stream
  HIP_TRY(c, hipMemsetAsync(c->prep_wf.p, 0xff, (size_t)B * 4, s));
  {
    Launch l(c, "k_proof_unpack");
    hipLaunchKernelGGL(k_proof_unpack, dim3(blocks_for((uint64_t)B * sh.proof_words, 256)), dim3(256), 0, s,
                       d_proofs, (uint64_t)proof_len, (uint32_t*)c->prep_pw.p, sh.proof_words, B,
                       (uint32_t*)c->prep_wf.p, proof_is_compact(sh, proof_len));
  }
  {
    Launch l(c, "k_transcript");
    hipLaunchKernelGGL(k_transcript, dim3(blocks_for(B, 64)), dim3(64), 0, s, sh, (const uint32_t*)plan->d_init,
                       (const uint4*)plan->d_tape, plan->n_ops, d_com, (const uint32_t*)c->prep_pw.p,
                       d_r, B, (uint32_t*)c->prep_ch.p, (uint32_t*)c->prep_wf.p, (const uint32_t*)plan->d_mono_chal,
                       (const uint32_t*)plan->d_mono_pow, 0u);
  }
  {
    Launch l(c, "k_prepare");
    hipLaunchKernelGGL(k_prepare, dim3(B), dim3(256), plan->lds_bytes, s, sh, (const uint32_t*)plan->d_mono_chal,
                       (const uint32_t*)plan->d_mono_pow, (const uint32_t*)plan->d_tgt_off, (const uint32_t*)plan->d_term_q,
                       (const uint2*)plan->d_term_mono, (const uint32_t*)plan->d_term_coef, (const uint32_t*)c->prep_ch.p,
                       d_com, (const uint32_t*)c->prep_pw.p, (uint32_t*)c->prep_dyn_sc.p,
                       (uint32_t*)c->recoded.p, (uint32_t*)c->prep_st_sc.p);
    hipLaunchKernelGGL(k_gather_dyn_points, dim3(blocks_for((uint64_t)B * sh.n_dyn * 8, 256)), dim3(256), 0, s, sh, d_com,
                       (const uint32_t*)c->prep_pw.p, B, (uint32_t*)c->prep_dyn_pt.p);
  }
  HIP_TRY(c, hipGetLastError());
  std::vector<uint8_t> bm((batch + 7) / 8, 0);
  int rc = ps->table ? batch_device_tables(c, job, ps, bm.data()) : batch_device(c, job, bm.data());
  park_sync_result(c, rc, bm.data(), batch);
  return ZKGPU_OK;
}

int cloak_verify_gpu_body(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch,
                          const uint32_t* d_com, const uint8_t* d_proofs, const uint32_t* d_r, size_t proof_len,
                          uint8_t* accept_bitmap) {
  TRY(cloak_verify_gpu_enqueue(c, ps, plan, batch, d_com, d_proofs, d_r, proof_len));
  int rc = pipe_wait(c, accept_bitmap);
  if (rc != ZKGPU_OK) memset(accept_bitmap, 0, (batch + 7) / 8);
  return rc;
}
}  // namespace

// Asynchronous form: enqueue the verification of one batch (inputs resident in HBM, as for
// zkgpu_cloak_verify_batch_gpu_dev) and return; zkgpu_verify_wait(ctx, bitmap) collects the result.
// One batch may be pending per context; fork contexts (zkgpu_ctx_fork) to keep several in flight.
int zkgpu_cloak_verify_submit_dev(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch,
                                  const void* d_commitments, const void* d_proofs, size_t proof_len, const void* d_r) {
  if (!c || !ps || !plan || plan->device != c->device || ps->ctx->device != c->device) return ZKGPU_EINVAL;
  if (batch == 0 || !d_commitments || !d_proofs || !d_r || batch >= (1ull << 24)) return ZKGPU_EINVAL;
  {
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (c->pending) return ZKGPU_EINVAL;
    if (!proof_len_fits(plan->shape, proof_len)) {     // wrong length for this statement: every proof is Err
      std::vector<uint8_t> z((batch + 7) / 8, 0);
      c->dep_event = nullptr;
      park_sync_result(c, ZKGPU_OK, z.data(), batch);
      return ZKGPU_OK;
    }
  }
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  return cloak_verify_gpu_enqueue(c, ps, plan, batch, (const uint32_t*)d_commitments, (const uint8_t*)d_proofs,
                                  (const uint32_t*)d_r, proof_len);
}

namespace {
// sizes the context's workspace for a whole-proof batch of `batch` statements of the plan's shape without launching anything
// (every buffer the pipeline would need: the ensure() calls of the enqueue path, run dry)
int cloak_reserve(zkgpu_ctx* c, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch, size_t proof_len) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (c->pending) return ZKGPU_EINVAL;
  DeviceGuard g(c->device);
  c->reserve_only = true;
  static const uint32_t dummy[4] = {0, 0, 0, 0};        // (never dereferenced: nothing is launched)
  const int rc = cloak_verify_gpu_enqueue(c, ps, plan, batch, dummy, (const uint8_t*)dummy, dummy, proof_len);
  c->reserve_only = false;
  c->last.valid = false;
  return rc;
}
}  // namespace

int zkgpu_verify_batch_ps_submit_dev(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t batch, const void* d_dyn_scalars,
                                     const void* d_dyn_points, const void* d_dyn_offsets, size_t n_dyn,
                                     const void* d_static_scalars, const void* d_static_index,
                                     const void* d_static_offsets, size_t n_static) {
  if (!c || !ps || ps->ctx->device != c->device || batch == 0 || batch >= (1ull << 31)) return ZKGPU_EINVAL;
  if (!d_dyn_offsets || !d_static_offsets) return ZKGPU_EINVAL;
  if ((n_dyn && (!d_dyn_scalars || !d_dyn_points)) || (n_static && !d_static_scalars)) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (c->pending) return ZKGPU_EINVAL;
  DeviceGuard g(c->device);
  Job job;
  job.d_dyn_scalars = (const uint32_t*)d_dyn_scalars;
  job.d_dyn_points = (const uint32_t*)d_dyn_points;
  job.d_dyn_offsets = (const uint64_t*)d_dyn_offsets;
  job.n_dyn = n_dyn;
  job.d_st_scalars = (const uint32_t*)d_static_scalars;
  job.d_st_index = (const uint32_t*)d_static_index;
  job.d_st_offsets = (const uint64_t*)d_static_offsets;
  job.n_static = n_static;
  job.d_static_rows = ps->rows;
  job.n_msm = (uint32_t)batch;
  if (pipe_eligible(c, job, ps)) return pipe_enqueue(c, job, ps, nullptr);
  std::vector<uint8_t> bm((batch + 7) / 8, 0);
  int rc = (ps->table && n_static) ? batch_device_tables(c, job, ps, bm.data()) : batch_device(c, job, bm.data());
  park_sync_result(c, rc, bm.data(), batch);
  return ZKGPU_OK;
}

int zkgpu_verify_wait(zkgpu_ctx* c, uint8_t* accept_bitmap) {
  if (!c || !accept_bitmap) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  const size_t nbytes = (c->pending_batch + 7) / 8;
  int rc = pipe_wait(c, accept_bitmap);
  if (rc != ZKGPU_OK && c->pending_batch) memset(accept_bitmap, 0, nbytes);
  return rc;
}

int zkgpu_cloak_verify_batch(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t gens_capacity, size_t batch,
                             const uint32_t* n_in, const uint32_t* n_out, const uint8_t* commitments,
                             const uint8_t* proofs, const uint64_t* proof_offsets, const uint8_t* r_bytes,
                             uint8_t* accept_bitmap, int host_threads) {
  if (!c || !ps || !accept_bitmap || (batch && (!n_in || !n_out || !commitments || !proofs || !proof_offsets)))
    return ZKGPU_EINVAL;
  memset(accept_bitmap, 0, (batch + 7) / 8);
  if (ps->n < 2 + 2 * gens_capacity) return ZKGPU_EINVAL;
  if (batch == 0) return ZKGPU_OK;
  CloakBatch cb;
  TRY(prepare_cloak_batch(gens_capacity, batch, n_in, n_out, commitments, proofs, proof_offsets, r_bytes,
                          host_threads, cb));
  int rc = zkgpu_verify_batch_ps(c, ps, batch, cb.dyn_sc.data(), cb.dyn_pt.data(), cb.dyn_off.data(), cb.st_sc.data(),
                                 cb.st_idx.data(), cb.st_off.data(), accept_bitmap);
  if (rc != ZKGPU_OK) { memset(accept_bitmap, 0, (batch + 7) / 8); return rc; }
  for (size_t i = 0; i < batch; ++i)
    if (!cb.wellformed[i]) accept_bitmap[i / 8] &= (uint8_t)~(1u << (i % 8));
  return ZKGPU_OK;
}

namespace {
// zkgpu_r1cs_desc -> R1csDesc (validated); false: malformed description
bool desc_from_c(zkgpu_ctx* c, const zkgpu_r1cs_desc* d, R1csDesc& desc) {
  if (!d || !d->transcript_label || (d->n_challenges && !d->challenge_labels) || (d->n_constraints && !d->term_offsets)) return false;
  if (d->n_multipliers_phase1 > d->n_multipliers || d->n_multipliers > (1u << 16) || d->n_commitments > (1u << 12)) return false;
  const uint64_t n_terms = d->n_constraints ? d->term_offsets[d->n_constraints] : 0;
  if (n_terms && (!d->term_var_kind || !d->term_var_index || !d->term_coeff || !d->term_challenge || !d->term_power)) return false;
  if (n_terms >= (1ull << 28)) return false;
  desc.label = d->transcript_label;
  desc.m = d->n_commitments; desc.n1 = d->n_multipliers_phase1; desc.n = d->n_multipliers;
  for (uint32_t i = 0; i < d->n_challenges; ++i) {
    if (!d->challenge_labels[i]) return false;
    desc.chal_names.push_back(d->challenge_labels[i]);
  }
  for (uint32_t q = 0; q < d->n_constraints; ++q) {
    if (d->term_offsets[q + 1] < d->term_offsets[q] || d->term_offsets[q + 1] > n_terms) return false;
    std::vector<R1csDesc::Term> con;
    for (uint64_t t = d->term_offsets[q]; t < d->term_offsets[q + 1]; ++t) {
      Scalar coef;
      if (d->term_var_kind[t] > 4 || !Scalar::from_canonical(d->term_coeff + 32 * t, coef)) { if (c) c->last_error = "r1cs description: bad variable kind or non-canonical coefficient"; return false; }
      if (d->term_challenge[t] >= (int32_t)d->n_challenges || d->term_power[t] > (1u << 20)) { if (c) c->last_error = "r1cs description: challenge index / power out of range"; return false; }
      con.push_back(R1csDesc::Term{(VarKind)d->term_var_kind[t], d->term_var_index[t], coef, d->term_challenge[t] < 0 ? -1 : d->term_challenge[t],
                                   d->term_challenge[t] < 0 ? 0u : d->term_power[t]});
    }
    desc.cons.push_back(std::move(con));
  }
  return true;
}
}  // namespace

// Host-prepared form for a described constraint system: transcript replay, flattening and scalars of every
// statement on host threads (r1cs_verifier.hpp through prepare_desc), the multiscalar multiplications in one
// device call.  Uniform batch: m x 32 bytes of commitments and proof_len bytes of proof per statement.
int zkgpu_r1cs_verify_batch(zkgpu_ctx* c, const zkgpu_pointset* ps, const zkgpu_r1cs_desc* d, size_t gens_capacity, size_t batch,
                            const uint8_t* commitments, const uint8_t* proofs, size_t proof_len, const uint8_t* r_bytes,
                            uint8_t* accept_bitmap, int host_threads) {
  if (!c || !ps || !accept_bitmap || !d || (batch && (!commitments || !proofs))) return ZKGPU_EINVAL;
  memset(accept_bitmap, 0, (batch + 7) / 8);
  if (ps->n < 2 + 2 * gens_capacity) return ZKGPU_EINVAL;
  R1csDesc desc;
  if (!desc_from_c(c, d, desc)) return ZKGPU_EINVAL;
  try { (void)plan_from_desc(desc); } catch (const std::exception& e) { c->last_error = e.what(); return ZKGPU_EINVAL; }
  if (batch == 0) return ZKGPU_OK;
  std::vector<uint8_t> rnd;
  if (!r_bytes) {
    rnd.resize(64 * batch);
    if (!os_random(rnd.data(), rnd.size())) return ZKGPU_EINVAL;
    r_bytes = rnd.data();
  }
  std::vector<VerifierMsm> prep(batch);
  std::vector<uint8_t> wellformed(batch, 0);
  const int nt = std::max(1, std::min<int>(host_threads > 0 ? host_threads : usable_cpus(), 256));
  auto work = [&](int tid) {
    for (size_t i = (size_t)tid; i < batch; i += (size_t)nt)
      wellformed[i] = prepare_desc(desc, commitments + 32ull * desc.m * i, proofs + proof_len * i, proof_len,
                                   Scalar::from_wide(r_bytes + 64 * i), gens_capacity, prep[i]) ? 1 : 0;
  };
  if (nt == 1) work(0);
  else { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work, t); for (auto& t : th) t.join(); }
  CloakBatch cb;
  cb.dyn_off.assign(batch + 1, 0); cb.st_off.assign(batch + 1, 0);
  for (size_t i = 0; i < batch; ++i) {
    cb.dyn_off[i + 1] = cb.dyn_off[i] + (wellformed[i] ? prep[i].dyn_scalars.size() / 32 : 0);
    cb.st_off[i + 1] = cb.st_off[i] + (wellformed[i] ? prep[i].static_scalars.size() / 32 : 0);
  }
  cb.dyn_sc.resize(32 * cb.dyn_off[batch]); cb.dyn_pt.resize(32 * cb.dyn_off[batch]);
  cb.st_sc.resize(32 * cb.st_off[batch]); cb.st_idx.resize(cb.st_off[batch]);
  for (size_t i = 0; i < batch; ++i) {
    if (!wellformed[i]) continue;
    memcpy(&cb.dyn_sc[32 * cb.dyn_off[i]], prep[i].dyn_scalars.data(), prep[i].dyn_scalars.size());
    memcpy(&cb.dyn_pt[32 * cb.dyn_off[i]], prep[i].dyn_points.data(), prep[i].dyn_points.size());
    memcpy(&cb.st_sc[32 * cb.st_off[i]], prep[i].static_scalars.data(), prep[i].static_scalars.size());
    memcpy(&cb.st_idx[cb.st_off[i]], prep[i].static_index.data(), prep[i].static_index.size() * 4);
  }
  int rc = zkgpu_verify_batch_ps(c, ps, batch, cb.dyn_sc.data(), cb.dyn_pt.data(), cb.dyn_off.data(), cb.st_sc.data(),
                                 cb.st_idx.data(), cb.st_off.data(), accept_bitmap);
  if (rc != ZKGPU_OK) { memset(accept_bitmap, 0, (batch + 7) / 8); return rc; }
  for (size_t i = 0; i < batch; ++i)
    if (!wellformed[i]) accept_bitmap[i / 8] &= (uint8_t)~(1u << (i % 8));
  return ZKGPU_OK;
}

int zkgpu_msm_batch(zkgpu_ctx* c, const uint8_t* scalars, const uint8_t* points, const uint64_t* offsets,
                    size_t batch, uint8_t* out, uint8_t* ok_bitmap) {
  if (!c || !out || !ok_bitmap || !offsets) return ZKGPU_EINVAL;
  memset(out, 0, 32 * batch);
  memset(ok_bitmap, 0, (batch + 7) / 8);
  uint64_t n = 0;
  if (!offsets_ok(offsets, batch, &n) || (n && (!scalars || !points)) || batch >= (1ull << 31)) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  TRY(upload(c, c->in_scalars, scalars, n * 32));
  TRY(upload(c, c->in_points, points, n * 32));
  TRY(upload(c, c->in_offsets, offsets, (batch + 1) * 8));
  Job job;
  job.d_dyn_scalars = (const uint32_t*)c->in_scalars.p;
  job.d_dyn_points = (const uint32_t*)c->in_points.p;
  job.d_dyn_offsets = (const uint64_t*)c->in_offsets.p;
  job.n_dyn = n;
  job.n_msm = (uint32_t)batch;
  job.max_dyn_row = longest_row(offsets, batch);
  int rc = batch_device(c, job, ok_bitmap, out);
  if (rc != ZKGPU_OK) { memset(out, 0, 32 * batch); memset(ok_bitmap, 0, (batch + 7) / 8); }
  return rc;
}

int zkgpu_hash_to_points(zkgpu_ctx* c, const uint8_t* uniform, size_t n, uint8_t* out) {
  if (!c || (n && (!uniform || !out))) return ZKGPU_EINVAL;
  if (n == 0) return ZKGPU_OK;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  TRY(upload(c, c->uniform, uniform, 64 * n));
  TRY(ensure(c, c->values, 32 * n));
  {
    Launch l(c, "k_from_uniform");
    hipLaunchKernelGGL(k_from_uniform, dim3(blocks_for(n, 64)), dim3(64), 0, c->stream,
                       (const uint32_t*)c->uniform.p, (uint32_t*)c->values.p, (uint64_t)n);
  }
  HIP_TRY(c, hipMemcpyAsync(out, c->values.p, 32 * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->profiling) prof_collect(c);
  return ZKGPU_OK;
}

// PedersenGens::default(): B = ristretto basepoint; B_blinding =
// from_uniform_bytes(SHA3-512(compress(B))).
int zkgpu_pedersen_gens(zkgpu_ctx* c, uint8_t B[32], uint8_t B_blinding[32]) {
  if (!c || !B || !B_blinding) return ZKGPU_EINVAL;
  ge base;
  base.X = fe_BASE_X(); base.Y = fe_BASE_Y(); base.Z = fe_one(); base.T = fe_BASE_T();
  uint32_t enc[8];
  ristretto_encode(enc, base);
  memcpy(B, enc, 32);
  uint8_t h[64];
  Sponge sp = sha3_512_sponge();
  sp.absorb(B, 32);
  sp.squeeze(h, 64);
  return zkgpu_hash_to_points(c, h, 1, B_blinding);
}

// BulletproofGens share `party`: G_i / H_i = from_uniform_bytes of consecutive
// 64-byte blocks of SHAKE256("GeneratorsChain" || 'G'/'H' || LE32(party)).
int zkgpu_bulletproof_gens(zkgpu_ctx* c, size_t capacity, uint32_t party, uint8_t* G, uint8_t* H) {
  if (!c || (capacity && (!G || !H))) return ZKGPU_EINVAL;
  std::vector<uint8_t> stream(64 * capacity);
  for (int side = 0; side < 2; ++side) {
    Sponge sp = shake256_sponge();
    const uint8_t label[5] = {(uint8_t)(side ? 'H' : 'G'), (uint8_t)party, (uint8_t)(party >> 8),
                              (uint8_t)(party >> 16), (uint8_t)(party >> 24)};
    sp.absorb((const uint8_t*)"GeneratorsChain", 15);
    sp.absorb(label, 5);
    sp.squeeze(stream.data(), stream.size());
    TRY(zkgpu_hash_to_points(c, stream.data(), capacity, side ? H : G));
  }
  return ZKGPU_OK;
}

// Diagnostics of the stream -> hardware-queue mapping (idle device assumed).  out[0]: do this context's two pipeline
// streams run side by side (1 / 0; -1: the probe failed) -- probed afresh; out[1]: GPU_MAX_HW_QUEUES as the process sees
// it (0: unset); out[2]: 1 when that value came too late -- the HIP runtime had started with the variable unset (seen by
// zkgpu_runtime_hint or by the first zkgpu_init of the process).  The variable is read by the HIP runtime ONCE, when it
// starts; the HOST exports it on zkgpu_runtime_hint's advice before its first HIP call -- the library never sets it -- and an
// application that touched HIP first runs on the runtime's default of 4 queues.  The library does not trust the variable: it
// probes what it got (here, and per lane in zkgpu_verifier_create).
int zkgpu_ctx_queue_info(zkgpu_ctx* c, int out[3]) {
  if (!c || !out) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  out[0] = (c->stream && c->stream2) ? streams_overlap(c->stream, c->stream2) : -1;
  const char* q = getenv("GPU_MAX_HW_QUEUES");
  out[1] = q ? atoi(q) : 0;
  out[2] = g_hw_queues_late ? 1 : 0;
  return ZKGPU_OK;
}

// Achievable HBM bandwidth by a streaming copy of `bytes` bytes (read + written: 2 x bytes per launch), best of `iters`
// launches timed with HIP events on the context's stream; *gbytes_per_s = 2 * bytes / time / 1e9.
int zkgpu_measure_hbm_copy(zkgpu_ctx* c, size_t bytes, int iters, double* gbytes_per_s) {
  if (!c || !gbytes_per_s || bytes < (1u << 20) || iters < 1 || iters > 1000) return ZKGPU_EINVAL;
  *gbytes_per_s = 0;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  bytes &= ~(size_t)15;
  void *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(&a, bytes);
  if (e == hipSuccess) e = hipMalloc(&b, bytes);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  if (e == hipSuccess) e = hipMemsetAsync(a, 0x5a, bytes, c->stream);
  float best = 0;
  bytes = std::min<size_t>(bytes, (size_t)0x7fffffffu * 4096);            // (one workgroup per 4 KiB: the grid's x dimension)
  const unsigned blocks = (unsigned)((bytes / 16 + 255) / 256);
  for (int it = 0; e == hipSuccess && it <= iters; ++it) {        // (the first launch warms up and is not counted)
    (void)hipEventRecord(e0, c->stream);
    hipLaunchKernelGGL(k_hbm_copy, dim3(blocks), dim3(256), 0, c->stream, (const uint4*)(it & 1 ? b : a), (uint4*)(it & 1 ? a : b), (uint64_t)(bytes / 16));
    (void)hipEventRecord(e1, c->stream);
    e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess && it > 0 && (best == 0 || ms < best)) best = ms;
  }
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (e != hipSuccess) { c->last_error = std::string("zkgpu_measure_hbm_copy: ") + hipGetErrorString(e); return e == hipErrorOutOfMemory ? ZKGPU_ENOMEM : ZKGPU_EHIP; }
  *gbytes_per_s = best > 0 ? 2.0 * (double)bytes / (best * 1e-3) / 1e9 : 0.0;
  return ZKGPU_OK;
}

int zkgpu_profile_enable(zkgpu_ctx* c, int on) {
  if (!c) return ZKGPU_EINVAL;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->profiling = on != 0;
  if (on && getenv("ZKGPU_TIMELINE")) {
    std::lock_guard<std::mutex> tl(g_timeline_mu);
    if (!g_timeline_ref) {
      DeviceGuard g(c->device);
      if (hipEventCreate(&g_timeline_ref) == hipSuccess) {
        (void)hipEventRecord(g_timeline_ref, c->stream_l);
        (void)hipEventSynchronize(g_timeline_ref);
      } else {
        g_timeline_ref = nullptr;
      }
    }
  }
  return ZKGPU_OK;
}

void zkgpu_profile_reset(zkgpu_ctx* c) {
  if (!c) return;
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  c->prof.clear();
}

int zkgpu_profile_count(zkgpu_ctx* c) { return c ? (int)c->prof.size() : 0; }

int zkgpu_profile_get(zkgpu_ctx* c, int i, const char** name, uint64_t* launches, double* total_ms) {
  if (!c || i < 0 || i >= (int)c->prof.size()) return ZKGPU_EINVAL;
  if (name) *name = c->prof[i].name;
  if (launches) *launches = c->prof[i].launches;
  if (total_ms) *total_ms = c->prof[i].ms;
  return ZKGPU_OK;
}

int zkgpu_last_window_bits(const zkgpu_ctx* c) { return c ? c->last_w : 0; }
uint64_t zkgpu_last_bucket_adds(const zkgpu_ctx* c) { return c ? c->last_adds : 0; }

int zkgpu_set_window_bits(zkgpu_ctx* c, int w) {
  if (!c || w < 0 || w > 16 || w == 1) return ZKGPU_EINVAL;
  c->forced_w = w;
  return ZKGPU_OK;
}

}  // extern "C"

#include "session.hpp"
