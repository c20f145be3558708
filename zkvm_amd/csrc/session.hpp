// session.hpp -- mixed-arity blocks of transactions and the multi-GPU exchange, behind the C ABI.
//
// Included at the end of zkgpu.hip (it uses the pipeline internals of that file).
//
//   zkgpu_verifier   mirror of the reference's `Verifier` for whole blocks: owns the contexts that keep
//                    several batches in flight and one device plan per statement shape.
//   zkgpu_txblock    a block of transactions of any mix of shapes, grouped by shape and resident in HBM
//                    (what `Tx::verify` would hand over for every transaction of a block).
//   zkgpu_comm       one RCCL communicator per process (= per GPU); the only exchange of the sharded
//                    verification is the all-gather of the per-shard accept bitmaps.
//
// RCCL is bound at run time (dlopen of librccl.so.1, the soname torch's bundled copy also carries), so
// the library loads and every single-GPU entry point works on a machine without it.
#pragma once
#include "comm_frame.hpp"
#include "tx_call.hpp"
#include "zkvm_tx.hpp"

#include <deque>
#include <dlfcn.h>
#include <rccl/rccl.h>

struct zkgpu_request {                 // one submitted batch (zkgpu_verifier_submit_dev)
  uint64_t id = 0;
  uint32_t n_in = 0, n_out = 0;
  size_t batch = 0, proof_len = 0, bit_off = 0;
  const void *d_com = nullptr, *d_proofs = nullptr, *d_r = nullptr;
  int state = 0;                       // 0 queued, 1 in flight on `lane`, 2 done
  int lane = -1, rc = 0;
  struct zkgpu_host_batch* form = nullptr;   // host-memory ticket: the device batch it was staged into (state 0 only)
  // a batch of a BLOCK (zkgpu_verifier_block_start): the run it belongs to (zkgpu_verifier::BlockRun*), which of the block's
  // shape groups, where in it; `ready`: the copy of the block to HBM is still queued -- whoever launches this waits for it
  void* run = nullptr;
  size_t group = 0, off = 0;
  hipEvent_t ready = nullptr;
  std::vector<uint8_t> bits;
};

// Tickets from HOST memory (zkgpu_verifier_submit): a device batch is FORMED in pinned host memory -- every ticket's
// commitments, proof bytes and randomness copied side by side at submission, so that the caller's buffers are free again
// when the call returns -- and goes to the device as three copies on the verifier's copy stream, straight into the lane's
// merge buffers: the merge that device tickets need as a kernel costs nothing here.
struct zkgpu_host_batch {
  uint32_t n_in = 0, n_out = 0;
  size_t proof_len = 0, wcom = 0;
  size_t cap_tx = 0, total = 0;        // transactions it has room for / holds
  size_t o_proofs = 0, o_r = 0;        // offsets of the three pieces inside the staging area
  int stage = -1;                      // which pinned area
  zkgpu_cloak_plan* plan = nullptr;
  std::vector<zkgpu_request*> members;
};

struct zkgpu_verifier {
  // ---- tickets: requests queued, merged by shape and run on the lanes (zkgpu_verifier_submit_dev / _wait)
  uint64_t next_id = 1;
  std::deque<zkgpu_request*> queue;                     // submitted, not launched
  std::map<uint64_t, zkgpu_request*> requests;          // every ticket not yet waited for
  std::vector<std::vector<zkgpu_request*>> running;     // per lane: the members of its merged batch in flight
  std::deque<int> busy;                                 // lanes in flight, oldest first
  size_t merge_target = 10240;                          // transactions per merged device batch of TICKETS: ten 1024-transaction batches,
                                                        // the arrangement every sweep of DESIGN.md sec 6 ends at (8192 ... 12 288 differ by
                                                        // 2 % in the steady state; below 8192 the chip-filling kernels lose their rounds)
  size_t block_merge = 4096;                            // the same for the batches of BLOCKS (zkgpu_verifier_verify_block, _block_start and
                                                        // the transaction calls): two lanes share a lone block of 8192 (measured, config 4)
  // host-memory tickets: pinned staging areas (2 * lanes + 2 of them, grow-only), each with a TWIN of the same size in HBM,
  // and the device batches being formed in them, oldest first; a batch that is full waits here for a free lane.  A ticket's
  // bytes travel to the twin as soon as they are staged (three hipMemcpyAsync per ticket on the copy stream), so that the copy
  // of a device batch runs beside the staging of its later tickets and a full batch leaves at once; the lane then reads the
  // twin in place -- the area stays taken until the batch has been collected (`lane_stage`).
  struct HostStage { void* pin = nullptr; void* dev = nullptr; size_t cap = 0; hipEvent_t copied = nullptr; bool taken = false; };
  std::vector<HostStage> host_stages;
  std::vector<int> lane_stage;                          // per lane: the staging area whose twin its batch in flight reads, or -1
  std::deque<std::unique_ptr<zkgpu_host_batch>> forming;
  zkgpu_ctx* root = nullptr;
  const zkgpu_pointset* ps = nullptr;
  size_t gens_capacity = 0;
  std::vector<zkgpu_ctx*> lanes;                      // root + forks: one batch in flight on each
  std::map<std::pair<uint32_t, uint32_t>, zkgpu_cloak_plan*> plans;   // nullptr: the shape cannot be verified here
  std::mutex plans_mu;                                  // (asked for by the staging thread of zkgpu_tx_verify_batch as well)
  std::map<std::pair<uint32_t, uint32_t>, uint64_t> costs;
  size_t chunk = 2048;                                // transactions per batch in flight
  // ---- blocks in flight (zkgpu_verifier_block_start / _finish): which batch of which block a lane is running
  // Since round 4 a block's batches ARE tickets: one request per (shape group, chunk) in the same queue, merged with the
  // requests of the same shape of OTHER blocks in flight into device batches of up to merge_target transactions.
  struct BlockRun {
    const struct zkgpu_txblock* b = nullptr;
    std::vector<uint8_t> bits;                          // verdicts gathered so far, by position in the block
    size_t pending = 0;                                 // batches of the block not yet collected (queued or on a lane)
    std::vector<zkgpu_request*> reqs;                   // those batches (the run owns them until they are collected)
    int rc = 0;
    uint64_t id = 0;
  };
  std::map<uint64_t, std::unique_ptr<BlockRun>> block_runs;
  uint64_t next_run = 1;
  int lanes_requested = 0, lanes_dropped = 0;           // lanes whose light stream shared a hardware queue with an earlier lane's were not kept
  int tx_format = 0;                                    // zkgpu_verifier_set_tx_format: 0 = no serialized-transaction format enabled
  // zkgpu_tx_verify_batch: two contexts of their own for the key and the signature stages (each a pair of streams beside
  // the lanes'), and a ring of staging areas (pinned host + device, grow-only) for the cloak statements of the chunks in
  // flight -- nothing on that path allocates or frees device memory once the sizes have been seen (hipFree synchronises)
  zkgpu_ctx* aux_keys[2] = {nullptr, nullptr};           // aggregated keys of the chunks, in turn
  zkgpu_ctx* aux_sigs[2] = {nullptr, nullptr};           // signature equations of the chunks, in turn: a stage may still run when the next is queued
  struct TxArena { void* h_pin = nullptr; size_t h_cap = 0; char* dev = nullptr; size_t d_cap = 0; hipEvent_t copied = nullptr; };
  hipStream_t copy_stream = nullptr;                    // staging copies run here; the lanes wait for TxArena::copied
  std::vector<TxArena> tx_arenas;
  std::vector<zk::zkvm::TxStatement> tx_statements;     // what the VM leaves per transaction, kept between calls (fresh memory costs
                                                        // a page fault per 4 KB: 0.4 ms per 3000 transactions, measured)
  std::vector<zk::zkvm::TxStatement> tx_statements_b;   // the same for the second round in flight (zkgpu_tx_verify_submit)
  size_t tx_statements_kept = (size_t)1 << 17;          // at most so many (~90 MB): zkgpu_verifier_set_tx_statements_kept
  size_t tx_chunk = 0;                                  // transactions per chunk of zkgpu_tx_verify_batch (0: automatic)
  uint8_t basepoint[32] = {0};                          // encoding of B, computed once (the signature equations name it)
  bool have_basepoint = false;
  std::mutex mu;
  std::string last_error;
  // zkgpu_tx_verify_submit / _wait: calls in flight.  An engine thread (made at the first submit) takes everything that is
  // queued and runs it as ONE merged call -- dynamic batching, as tickets do for proofs -- then hands every caller its own
  // bits.  tx_mu guards the queue and the records; the engine takes `mu` for the length of a round like any other call.
  struct TxPending {
    uint64_t id = 0;
    size_t batch = 0;
    const uint8_t* txs = nullptr;
    const uint64_t* offsets = nullptr;
    int host_threads = 0;
    int state = 0, rc = 0;                              // 0 queued, 1 in a round, 2 done
    std::chrono::steady_clock::time_point arrived;      // (the engine lets a small queue wait a little while a round is running)
    std::vector<uint8_t> bits, status;
  };
  std::mutex tx_mu;
  std::condition_variable tx_cv;
  std::deque<TxPending*> tx_queue;
  std::map<uint64_t, std::unique_ptr<TxPending>> tx_calls;
  uint64_t tx_next_id = 1;
  std::thread tx_engine;
  bool tx_engine_quit = false;
  size_t tx_merge_max = 16384;                          // transactions per merged round
  uint64_t tx_rounds = 0, tx_round_calls = 0;           // statistics: rounds run, calls they held (zkgpu_tx_verify_stats)
};

struct zkgpu_txblock {
  zkgpu_verifier* v = nullptr;
  size_t batch = 0;
  bool owns_dev = true;                                 // false: `dev` is a staging area of the verifier (tx path)
  hipEvent_t ready = nullptr;                           // set: the copy to `dev` is still queued; batches wait for this event
  struct Group {
    uint32_t n_in, n_out;
    size_t proof_len;
    zkgpu_cloak_plan* plan;                           // nullptr: every transaction of the group is rejected
    std::vector<uint32_t> idx;                        // positions in the block, in order
    size_t com_off, proof_off, r_off;                 // byte offsets into `dev`
  };
  std::vector<Group> groups;
  char* dev = nullptr;                                // one allocation: commitments | proofs | r, group after group
  size_t dev_bytes = 0;
  std::unique_ptr<uint8_t[]> host_owned;              // staged image between txblock_stage_host and txblock_upload (no arena)
  const uint8_t* host_image = nullptr;
};

struct zkgpu_comm {
  zkgpu_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  const void* api = nullptr;                            // RcclApi*: the function table this communicator was made with, for its whole life
  hipStream_t stream = nullptr;
  // exchange buffers of a fixed size, made with the communicator: no allocation -- nothing that can fail on ONE rank --
  // stands between a call and its collective (slot = bytes one rank contributes, at most COMM_MAX_SLOT)
  void* d_send = nullptr; void* d_recv = nullptr;
  void* h_pin = nullptr;                                // [slot | world * slot]
  std::mutex mu;
  std::string last_error;
};
constexpr size_t COMM_MAX_SLOT = 1u << 20;             // bytes per rank and call: a status word + the bitmap of 8 M transactions
constexpr uint32_t COMM_POISON = 0x80000001u;          // what d_send's first word holds between calls: "this rank failed"

namespace {
// verifiers alive per device in this process (zkgpu_verifier_create warns about a second one: ZKGPU_WSECOND_VERIFIER)
std::mutex g_live_verifiers_mu;
std::map<int, int> g_live_verifiers;

void ticket_collect(zkgpu_verifier* v, int lane);
int ticket_dispatch(zkgpu_verifier* v, bool force);
int host_dispatch(zkgpu_verifier* v, zkgpu_host_batch* must);
void collect_oldest(zkgpu_verifier* v);

// ---- RCCL, bound on first use -------------------------------------------------------------
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
  bool ok() const { return handle && GetUniqueId && CommInitRank && AllGather && CommDestroy && GetErrorString; }
};

// Test hook (zkgpu_debug_comm_mock): an in-process stand-in for a world of N ranks behind the same function table, so
// that the exchange step -- buffers, stream, framing, fail-closed statuses, the poison word -- can run at world 2 .. 8 on
// ONE GPU (RCCL refuses two ranks on one device).  The mock's all-gather puts this rank's bytes at its place and fills the
// other ranks' places from the slots the test supplied.
struct CommMock {
  bool on = false;
  int world = 1, rank = 0;
  size_t slot = 0;
  std::vector<uint8_t> peers;                     // world x slot bytes
  void* d_peers = nullptr;
  uint64_t gathers = 0;
};
CommMock g_comm_mock;
std::mutex g_comm_mock_mu;

ncclResult_t mock_get_unique_id(ncclUniqueId* id) { memset(id, 0x5a, sizeof *id); return ncclSuccess; }
ncclResult_t mock_comm_init_rank(ncclComm_t* comm, int world, ncclUniqueId, int rank) {
  std::lock_guard<std::mutex> lk(g_comm_mock_mu);
  if (world != g_comm_mock.world) return ncclInvalidArgument;
  g_comm_mock.rank = rank;
  *comm = (ncclComm_t)&g_comm_mock;
  return ncclSuccess;
}
ncclResult_t mock_all_gather(const void* send, void* recv, size_t bytes, ncclDataType_t, ncclComm_t, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_comm_mock_mu);
  CommMock& m = g_comm_mock;
  ++m.gathers;
  zk::fault::Suppress outside_the_product;
  if (bytes != m.slot) return ncclInvalidArgument;
  if (hipMemcpyAsync(recv, m.d_peers, bytes * (size_t)m.world, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpyAsync((char*)recv + bytes * (size_t)m.rank, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}
ncclResult_t mock_comm_destroy(ncclComm_t) { return ncclSuccess; }
const char* mock_error_string(ncclResult_t) { return "mock collective error"; }

// The function table a communicator uses is chosen ONCE, when it is created, and stays with it (zkgpu_comm::api): the
// switch of the test hook is read under its lock at that moment only, so toggling it later cannot route a live
// communicator's all-gather or its destruction to the other table.
bool test_hooks_enabled();
// Test hook for the bring-up rehearsal (bench.py's bounded communicator bring-up, tests/test_launch.py): with
// ZKGPU_TEST_HOOKS=1 AND ZKGPU_TEST_COMM_STALL="init:<rank>|init:all|gather:<rank>|gather:all" in the environment when
// RCCL is first bound, ncclCommInitRank (or the first ncclAllGather) of the named rank never returns -- the failure RCCL
// shows when one rank of a node cannot reach the others, which a process cannot cancel from inside.  Inert otherwise.
struct CommStall {
  int where = 0, rank = -1, my_rank = -1;       // where: 1 init, 2 gather; rank -1: every rank
  ncclResult_t (*init)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
};
CommStall g_comm_stall;
[[noreturn]] void stall_for_ever(const char* what) {
  fprintf(stderr, "[zkgpu test hook] %s stalls (ZKGPU_TEST_COMM_STALL)\n", what);
  for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
}
ncclResult_t stall_comm_init_rank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
  g_comm_stall.my_rank = rank;
  if (g_comm_stall.where == 1 && (g_comm_stall.rank < 0 || g_comm_stall.rank == rank)) stall_for_ever("ncclCommInitRank");
  return g_comm_stall.init(comm, world, id, rank);
}
ncclResult_t stall_all_gather(const void* s, void* r, size_t n, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
  if (g_comm_stall.where == 2 && (g_comm_stall.rank < 0 || g_comm_stall.rank == g_comm_stall.my_rank)) stall_for_ever("ncclAllGather");
  return g_comm_stall.gather(s, r, n, t, c, st);
}
RcclApi& rccl_real() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.handle) break;
    }
    if (!api.handle) { api.error = std::string("RCCL not found: ") + dlerror(); return; }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
    api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
    if (!api.ok()) { api.error = "RCCL library lacks an expected symbol"; return; }
    const char* st = test_hooks_enabled() ? getenv("ZKGPU_TEST_COMM_STALL") : nullptr;
    if (st && (!strncmp(st, "init:", 5) || !strncmp(st, "gather:", 7))) {
      CommStall& cs = g_comm_stall;
      cs.where = st[0] == 'i' ? 1 : 2;
      const char* who = strchr(st, ':') + 1;
      cs.rank = !strcmp(who, "all") ? -1 : atoi(who);
      cs.init = api.CommInitRank; cs.gather = api.AllGather;
      api.CommInitRank = stall_comm_init_rank; api.AllGather = stall_all_gather;
    }
  });
  return api;
}
RcclApi& rccl_mock() {
  static RcclApi mock;
  static std::once_flag once;
  std::call_once(once, [] {
    mock.handle = (void*)&g_comm_mock;
    mock.GetUniqueId = mock_get_unique_id; mock.CommInitRank = mock_comm_init_rank; mock.AllGather = mock_all_gather;
    mock.CommDestroy = mock_comm_destroy; mock.GetErrorString = mock_error_string;
  });
  return mock;
}
const RcclApi& comm_api(const zkgpu_comm* cm) { return *(const RcclApi*)cm->api; }
RcclApi& rccl_for_new_comm() {
  bool mocked;
  { std::lock_guard<std::mutex> lk(g_comm_mock_mu); mocked = g_comm_mock.on; }
  return mocked ? rccl_mock() : rccl_real();
}
// The hook exists in the shipped library only for processes that ask for it BEFORE they load it: a deployed verifier never
// has the variable set, and zkgpu_debug_comm_mock refuses.
bool test_hooks_enabled() {
  static const bool on = [] { const char* e = getenv("ZKGPU_TEST_HOOKS"); return e && e[0] == '1'; }();
  return on;
}

// number of terms of the verification multiscalar multiplication of one cloak statement: the weight
// by which a mixed block is balanced over the GPUs (0: not a provable shape)
uint64_t cloak_msm_terms(uint32_t n_in, uint32_t n_out) {
  if (n_in + n_out == 0 || n_in > 64 || n_out > 64) return 0;
  try {
    const CloakPlan p = PlanBuilder::build(n_in, n_out);
    return 11ull + p.m + 2ull * p.k + 2 + 2ull * p.pn;
  } catch (const std::exception&) {
    return 0;
  }
}

// plan of a shape, created on first use.  *rc = ZKGPU_OK and nullptr: a shape the generator set can NEVER serve (more
// multipliers than generators, no values, more than 64 inputs / outputs) -- the reference rejects exactly those
// transactions (InvalidGeneratorsLength / a VM error) and so does the caller, one by one; that answer is cached.
// *rc != ZKGPU_OK: the plan could not be made THIS time (out of device memory, a HIP error): nothing is cached, and the
// caller fails its block or ticket with that error -- a transient fault must not turn into "the proof is invalid".
// May run on the staging thread of zkgpu_tx_verify_batch: the error text goes to `err` (the caller's, who knows which lock
// guards v->last_error), never to the verifier from here.
zkgpu_cloak_plan* verifier_plan(zkgpu_verifier* v, uint32_t n_in, uint32_t n_out, int* rc, std::string* err) {
  *rc = ZKGPU_OK;
  std::lock_guard<std::mutex> plk(v->plans_mu);
  const auto key = std::make_pair(n_in, n_out);
  auto it = v->plans.find(key);
  if (it != v->plans.end()) return it->second;
  zkgpu_cloak_plan* p = nullptr;
  const int r = zkgpu_cloak_plan_create(v->root, n_in, n_out, v->gens_capacity, &p);
  if (r != ZKGPU_OK && r != ZKGPU_EINVAL) { *rc = r; if (err) *err = zkgpu_last_error(v->root); return nullptr; }
  if (r != ZKGPU_OK) p = nullptr;
  v->plans[key] = p;
  return p;
}

int drain(zkgpu_verifier* v, std::vector<uint8_t>& scratch) {
  int first = ZKGPU_OK;
  for (zkgpu_ctx* c : v->lanes) {
    bool pending;
    size_t nb;
    { std::lock_guard<std::recursive_mutex> lk(c->mu); pending = c->pending; nb = (c->pending_batch + 7) / 8; }
    if (!pending) continue;
    scratch.resize(std::max<size_t>(nb, 1));
    const int rc = zkgpu_verify_wait(c, scratch.data());
    if (rc != ZKGPU_OK && first == ZKGPU_OK) first = rc;
  }
  return first;
}

}  // namespace

extern "C" {

int zkgpu_verifier_create(zkgpu_ctx* ctx, const zkgpu_pointset* ps, size_t gens_capacity, int batches_in_flight,
                          zkgpu_verifier** out) {
  if (!ctx || !ps || !out || ps->ctx->device != ctx->device || ps->n < 2 + 2 * gens_capacity) return ZKGPU_EINVAL;
  *out = nullptr;
  if (batches_in_flight <= 0) batches_in_flight = 6;
  batches_in_flight = std::min(batches_in_flight, 1 + MAX_FORKS);
  zkgpu_verifier* v = new zkgpu_verifier();
  v->root = ctx; v->ps = ps; v->gens_capacity = gens_capacity;
  v->lanes.push_back(ctx);
  v->lanes_requested = batches_in_flight;
  // A lane is worth having only if its light stream -- the latency-bound kernels of its batch: transcript, Horner chains,
  // verdicts -- really runs beside the other lanes'.  The runtime hands out a limited number of hardware queues
  // (GPU_MAX_HW_QUEUES, read once when HIP starts: 4 unless the process exported more BEFORE its first HIP call -- the host does,
  // on zkgpu_runtime_hint's advice), and streams beyond that share queues: two lanes on one queue take turns, and when they
  // also wait for each other's events they crawl (measured in round 2: 3x - 10x slower with 8 and 10 lanes than with 7 and
  // 9).  So every new lane is probed against the lanes kept so far (two spinning wavefronts, ~0.2 ms per pair, idle
  // device assumed) and not kept if it serialises with one of them; a stream made next lands on the runtime's next queue, so
  // a rejected lane is tried again, up to as many times as lanes were asked for: fewer lanes in the end, every one real.
  {
    DeviceGuard g(ctx->device);
    int retries = batches_in_flight;
    while ((int)v->lanes.size() < batches_in_flight) {
      zkgpu_ctx* f = nullptr;
      if (zkgpu_ctx_fork(ctx, &f) != ZKGPU_OK) break;     // fewer lanes than asked for: still correct
      bool alone = true;
      for (zkgpu_ctx* kept : v->lanes)
        if (streams_overlap(kept->stream_l, f->stream_l) == 0) { alone = false; break; }
      if (alone) { v->lanes.push_back(f); continue; }
      zkgpu_destroy(f);
      if (retries-- <= 0) break;
    }
    v->lanes_dropped = batches_in_flight - (int)v->lanes.size();
  }
  if (g_hw_queues_late) {
    v->last_error = "the HIP runtime of this process started before GPU_MAX_HW_QUEUES was set: it runs on its default of 4 hardware "
                    "queues, on which batches in flight take turns -- export GPU_MAX_HW_QUEUES=18 before the process's first HIP call";
  } else if (v->lanes_dropped) {
    char msg[256];
    snprintf(msg, sizeof msg, "%d of %d lanes not kept: their streams share a hardware queue with another lane's (GPU_MAX_HW_QUEUES=%s; "
             "export it before the process's first HIP call)", v->lanes_dropped, batches_in_flight, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "unset");
    v->last_error = msg;
  }
  v->running.resize(v->lanes.size());
  *out = v;
  // One verifier per process and device is what the queue budget is made for: a second one is created
  // all the same, and the caller is TOLD when the runtime's queue count is in the range where two verifiers were measured to
  // stall each other (20 and more) -- a positive status, the only one the library has.
  int others;
  { std::lock_guard<std::mutex> lk(g_live_verifiers_mu); others = g_live_verifiers[ctx->device]++; }
  const char* q = getenv("GPU_MAX_HW_QUEUES");
  if (others > 0 && !g_hw_queues_late && q && atoi(q) >= 20) {
    char msg[320];
    snprintf(msg, sizeof msg, "%d other verifier(s) alive on device %d with GPU_MAX_HW_QUEUES=%s: the streams of two verifiers oversubscribe the "
             "device's queue slots (calls of 5 - 40 ms instead of a steady 6): use one verifier per process and device, or export "
             "GPU_MAX_HW_QUEUES=16 before the first HIP call", others, ctx->device, q);
    v->last_error = msg;
    return ZKGPU_WSECOND_VERIFIER;
  }
  return ZKGPU_OK;
}

void zkgpu_verifier_destroy(zkgpu_verifier* v) {
  if (!v) return;
  { std::lock_guard<std::mutex> lk(g_live_verifiers_mu); --g_live_verifiers[v->root->device]; }
  {                                                      // the engine of the transaction calls in flight: told, and waited for
    { std::lock_guard<std::mutex> lk(v->tx_mu); v->tx_engine_quit = true; }
    v->tx_cv.notify_all();
    if (v->tx_engine.joinable()) v->tx_engine.join();
  }
  std::vector<uint8_t> scratch;
  (void)drain(v, scratch);
  for (auto& kv : v->requests) delete kv.second;
  for (auto& kv : v->block_runs) for (zkgpu_request* r : kv.second->reqs) delete r;    // (runs never finished: queued batches)
  for (size_t i = 1; i < v->lanes.size(); ++i) zkgpu_destroy(v->lanes[i]);
  for (zkgpu_ctx* a : v->aux_keys) if (a) zkgpu_destroy(a);
  for (zkgpu_ctx* a : v->aux_sigs) if (a) zkgpu_destroy(a);
  {
    DeviceGuard g(v->root->device);
    for (auto& a : v->tx_arenas) { if (a.h_pin) (void)hipHostFree(a.h_pin); if (a.dev) (void)hipFree(a.dev); if (a.copied) (void)hipEventDestroy(a.copied); }
    for (auto& hs : v->host_stages) { if (hs.pin) (void)hipHostFree(hs.pin); if (hs.dev) (void)hipFree(hs.dev); if (hs.copied) (void)hipEventDestroy(hs.copied); }
    if (v->copy_stream) (void)hipStreamDestroy(v->copy_stream);
  }
  for (auto& kv : v->plans) if (kv.second) zkgpu_cloak_plan_destroy(kv.second);
  delete v;
}

int zkgpu_verifier_set_chunk(zkgpu_verifier* v, size_t transactions) {
  if (!v || transactions == 0 || transactions >= (1u << 24)) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  v->chunk = transactions;
  return ZKGPU_OK;
}

int zkgpu_verifier_lanes(const zkgpu_verifier* v) { return v ? (int)v->lanes.size() : 0; }

// out[0] lanes in use, out[1] lanes asked for, out[2] lanes dropped at creation because their stream did not run beside
// another lane's, out[3] 1 when the process's HIP runtime started before GPU_MAX_HW_QUEUES was set (zkgpu_verifier_last_error
// says which)
int zkgpu_verifier_queue_info(const zkgpu_verifier* v, int out[4]) {
  if (!v || !out) return ZKGPU_EINVAL;
  out[0] = (int)v->lanes.size(); out[1] = v->lanes_requested; out[2] = v->lanes_dropped; out[3] = g_hw_queues_late ? 1 : 0;
  return ZKGPU_OK;
}

// the context of lane i (0 = the one the verifier was created on): for the measurement hooks
// (zkgpu_profile_*, zkgpu_set_serial, zkgpu_set_group_size); owned by the verifier
zkgpu_ctx* zkgpu_verifier_lane(zkgpu_verifier* v, int i) {
  return (v && i >= 0 && i < (int)v->lanes.size()) ? v->lanes[(size_t)i] : nullptr;
}

// The text is COPIED under the verifier's mutex into a buffer of the calling thread (valid until that thread asks again):
// calls on one verifier may come from many threads (the Rust wrapper is Sync), and a pointer into the shared string
// could be read while another thread's failing call rewrites it.
const char* zkgpu_verifier_last_error(const zkgpu_verifier* v) {
  if (!v) return "";
  static thread_local std::string mine;
  std::lock_guard<std::mutex> lk(const_cast<zkgpu_verifier*>(v)->mu);
  mine = v->last_error;
  return mine.c_str();
}

uint64_t zkgpu_cloak_msm_terms(uint32_t n_in, uint32_t n_out) { return cloak_msm_terms(n_in, n_out); }

void zkgpu_txblock_destroy(zkgpu_txblock* b) {
  if (!b) return;
  if (b->dev && b->owns_dev) { DeviceGuard g(b->v->root->device); (void)hipFree(b->dev); }
  delete b;
}

size_t zkgpu_txblock_size(const zkgpu_txblock* b) { return b ? b->batch : 0; }
size_t zkgpu_txblock_shapes(const zkgpu_txblock* b) { return b ? b->groups.size() : 0; }

namespace {

// One transaction of a block being staged: where its commitments (64 (n_in + n_out) bytes) and its proof lie in host memory
using TxSource = zk::zkvm::TxProofSource;

// groups the transactions by (inputs, outputs, proof length), lays the groups out and gathers them on host threads into
// the staging image (the arena's pinned memory, or a buffer of the block's own): host work only -- apart from the first
// sight of a shape (its plan) and of a size (the arena), which touch the device.  `err` receives what went wrong.
// May run on another thread than the one that owns the verifier (zkgpu_tx_verify_batch): touches the plans (plans_mu),
// the given arena and nothing else of *v.
int txblock_stage_host(zkgpu_verifier* v, size_t batch, const TxSource* src, const uint8_t* r_bytes, int host_threads, zkgpu_txblock** out,
                       zkgpu_verifier::TxArena* arena, std::string* err) {
  zkgpu_ctx* c = v->root;
  std::unique_ptr<zkgpu_txblock> b(new zkgpu_txblock());
  b->v = v; b->batch = batch;
  std::map<std::tuple<uint32_t, uint32_t, uint64_t>, size_t> where;
  std::vector<uint32_t> slot(batch);                  // position inside its group
  std::vector<uint32_t> group_of(batch);
  size_t last = (size_t)-1;
  std::tuple<uint32_t, uint32_t, uint64_t> last_key{0, 0, 0};
  for (size_t i = 0; i < batch; ++i) {
    const auto key = std::make_tuple(src[i].n_in, src[i].n_out, src[i].proof_len);
    if (last == (size_t)-1 || key != last_key) {      // blocks are mostly runs of one shape: the map is asked once per run
      auto it = where.find(key);
      if (it == where.end()) {
        zkgpu_txblock::Group g;
        g.n_in = src[i].n_in; g.n_out = src[i].n_out; g.proof_len = (size_t)src[i].proof_len;
        int prc = ZKGPU_OK;
        g.plan = verifier_plan(v, g.n_in, g.n_out, &prc, err);
        if (prc != ZKGPU_OK) return prc;
        if (g.plan && !proof_len_fits(g.plan->shape, g.proof_len)) g.plan = nullptr;   // wrong length for the statement
        g.com_off = g.proof_off = g.r_off = 0;
        it = where.emplace(key, b->groups.size()).first;
        b->groups.push_back(std::move(g));
      }
      last = it->second;
      last_key = key;
    }
    group_of[i] = (uint32_t)last;
    slot[i] = (uint32_t)b->groups[last].idx.size();
    b->groups[last].idx.push_back((uint32_t)i);
  }
  // layout in HBM (256-byte aligned pieces), only for the groups that will run
  size_t total = 0;
  auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
  for (auto& g : b->groups) {
    if (!g.plan) continue;
    const size_t n = g.idx.size(), wcom = 64 * ((size_t)g.n_in + g.n_out);
    g.com_off = total; total = align(total + n * wcom);
    g.proof_off = total; total = align(total + n * g.proof_len);
    g.r_off = total; total = align(total + n * 64);
  }
  b->dev_bytes = total;
  if (total) {
    uint8_t* host = nullptr;
    if (arena) {                                        // the verifier's staging area: pinned host memory, device memory kept
      DeviceGuard dg(c->device);
      if (arena->h_cap < total) {
        if (arena->h_pin) (void)hipHostFree(arena->h_pin);
        arena->h_pin = nullptr; arena->h_cap = 0;
        const size_t want = total + total / 4 + 4096;
        if (hipHostMalloc(&arena->h_pin, want, hipHostMallocDefault) != hipSuccess) { *err = "hipHostMalloc (transaction staging)"; return ZKGPU_ENOMEM; }
        arena->h_cap = want;
      }
      if (arena->d_cap < total) {
        if (arena->dev) (void)hipFree(arena->dev);
        arena->dev = nullptr; arena->d_cap = 0;
        const size_t want = total + total / 4 + 4096;
        if (hipMalloc((void**)&arena->dev, want) != hipSuccess) { *err = "hipMalloc (transaction staging)"; return ZKGPU_ENOMEM; }
        arena->d_cap = want;
      }
      host = (uint8_t*)arena->h_pin;
    } else {
      b->host_owned.reset(new uint8_t[total]);
      host = b->host_owned.get();
    }
    b->host_image = host;
    // verifier randomness when the caller gives none: 32 bytes from the OS (getrandom(2)) per block, expanded per
    // transaction with SHAKE256(seed || position) on the gathering threads -- getrandom itself delivers ~0.35 GB/s on one
    // thread, which for 64 bytes per transaction would be a quarter of this stage's time
    uint8_t seed[40] = {0};
    if (!r_bytes && !os_random(seed, 32)) { *err = "getrandom failed"; return ZKGPU_EINVAL; }
    host_parallel(batch, host_threads, [&](size_t i) {
      const zkgpu_txblock::Group& g = b->groups[group_of[i]];
      if (!g.plan) return;
      const size_t wcom = 64 * ((size_t)g.n_in + g.n_out), j = slot[i];
      memcpy(&host[g.com_off + j * wcom], src[i].com, wcom);
      memcpy(&host[g.proof_off + j * g.proof_len], src[i].proof, g.proof_len);
      if (r_bytes) {
        memcpy(&host[g.r_off + j * 64], r_bytes + 64 * i, 64);
      } else {
        uint8_t in[40];
        memcpy(in, seed, 32);
        for (int q = 0; q < 8; ++q) in[32 + q] = (uint8_t)((uint64_t)i >> (8 * q));
        Sponge sp = shake256_sponge();
        sp.absorb(in, 40);
        sp.squeeze(&host[g.r_off + j * 64], 64);
      }
    });
  }
  *out = b.release();
  return ZKGPU_OK;
}

// the staged image -> HBM: queued on the copy stream when the block lives in an arena (the host goes on, the lanes wait
// for the event), a plain copy into memory of the block's own otherwise.  (v->mu held.)
int txblock_upload(zkgpu_verifier* v, zkgpu_txblock* b, zkgpu_verifier::TxArena* arena) {
  const size_t total = b->dev_bytes;
  if (!total) return ZKGPU_OK;
  DeviceGuard dg(v->root->device);
  hipError_t e;
  if (arena) {
    b->dev = arena->dev;
    b->owns_dev = false;
    e = v->copy_stream ? hipSuccess : hipStreamCreateWithFlags(&v->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess && !arena->copied) e = hipEventCreateWithFlags(&arena->copied, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMemcpyAsync(b->dev, b->host_image, total, hipMemcpyHostToDevice, v->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(arena->copied, v->copy_stream);
    if (e == hipSuccess) b->ready = arena->copied;
  } else {
    e = hipMalloc((void**)&b->dev, total);
    if (e != hipSuccess) { v->last_error = std::string("hipMalloc: ") + hipGetErrorString(e); b->dev = nullptr; return ZKGPU_ENOMEM; }
    e = hipMemcpy(b->dev, b->host_image, total, hipMemcpyHostToDevice);
    b->host_owned.reset();
    b->host_image = nullptr;
  }
  if (e != hipSuccess) {
    v->last_error = hipGetErrorString(e);
    quiesce_after_fault(v->root);        // (an arena's copy may be half queued: the arena is handed out again after this)
    if (b->owns_dev && b->dev) (void)hipFree(b->dev);
    b->dev = nullptr;
    return ZKGPU_EHIP;
  }
  return ZKGPU_OK;
}

// (v->mu held: last_error is the verifier's)
int txblock_build_locked(zkgpu_verifier* v, size_t batch, const TxSource* src, const uint8_t* r_bytes, int host_threads, zkgpu_txblock** out,
                         zkgpu_verifier::TxArena* arena = nullptr) {
  std::string err;
  zkgpu_txblock* b = nullptr;
  int rc = txblock_stage_host(v, batch, src, r_bytes, host_threads, &b, arena, &err);
  if (rc != ZKGPU_OK) { v->last_error = err; return rc; }
  rc = txblock_upload(v, b, arena);
  if (rc != ZKGPU_OK) { zkgpu_txblock_destroy(b); return rc; }
  *out = b;
  return ZKGPU_OK;
}

int txblock_build(zkgpu_verifier* v, size_t batch, const TxSource* src, const uint8_t* r_bytes, int host_threads, zkgpu_txblock** out) {
  std::lock_guard<std::mutex> lk(v->mu);
  return txblock_build_locked(v, batch, src, r_bytes, host_threads, out);
}

}  // namespace

int zkgpu_txblock_create(zkgpu_verifier* v, size_t batch, const uint32_t* n_in, const uint32_t* n_out,
                         const uint8_t* commitments, const uint8_t* proofs, const uint64_t* proof_offsets,
                         const uint8_t* r_bytes, zkgpu_txblock** out) {
  if (!v || !out) return ZKGPU_EINVAL;
  *out = nullptr;
  if (batch && (!n_in || !n_out || !commitments || !proofs || !proof_offsets)) return ZKGPU_EINVAL;
  if (batch >= (1ull << 31)) return ZKGPU_EINVAL;
  for (size_t i = 0; i < batch; ++i) if (proof_offsets[i + 1] < proof_offsets[i]) return ZKGPU_EINVAL;
  std::vector<TxSource> src(batch);
  uint64_t com_off = 0;
  for (size_t i = 0; i < batch; ++i) {
    src[i] = TxSource{n_in[i], n_out[i], commitments + com_off, proofs + proof_offsets[i], proof_offsets[i + 1] - proof_offsets[i]};
    com_off += 64ull * ((uint64_t)n_in[i] + n_out[i]);
  }
  return txblock_build(v, batch, src.data(), r_bytes, 0, out);
}

// Verifies every transaction of a resident block: the groups are cut into batches of at most `chunk`
// transactions, which go round the verifier's lanes (one batch in flight on each); bit i of
// accept_bitmap is the verdict of transaction i of the block.  Any device error: all bits zero.
namespace {

// Queues every batch of the block as a request of the ticket queue (v->mu held).  now: launch at once whatever the merge
// target says (a block somebody is about to wait for); otherwise the requests wait -- for requests of the same shape from
// the next blocks, until the merge target is reached -- and block_finish forces them out.
// nullptr: out of host memory while queueing (nothing of the block is left queued; v->last_error says so).
zkgpu_verifier::BlockRun* block_start(zkgpu_verifier* v, const zkgpu_txblock* b, bool now) {
  zkgpu_verifier::BlockRun* run = nullptr;
  try {
    std::unique_ptr<zkgpu_verifier::BlockRun> owned(new zkgpu_verifier::BlockRun());
    run = owned.get();
    run->b = b;
    run->bits.assign((b->batch + 7) / 8, 0);
    run->id = v->next_run++;
    v->block_runs[run->id] = std::move(owned);
    for (size_t gi = 0; gi < b->groups.size(); ++gi) {
      const auto& g = b->groups[gi];
      if (!g.plan) continue;
      const size_t wcom = 64 * ((size_t)g.n_in + g.n_out);
      for (size_t off = 0; off < g.idx.size(); off += v->chunk) {
        std::unique_ptr<zkgpu_request> r(new zkgpu_request());
        r->id = 0;                                        // (not a ticket of the caller's: found through its run)
        r->n_in = g.n_in; r->n_out = g.n_out; r->proof_len = g.proof_len;
        r->batch = std::min(v->chunk, g.idx.size() - off);
        r->d_com = b->dev + g.com_off + off * wcom;
        r->d_proofs = b->dev + g.proof_off + off * g.proof_len;
        r->d_r = b->dev + g.r_off + off * 64;
        r->run = run; r->group = gi; r->off = off; r->ready = b->ready;
        run->reqs.reserve(run->reqs.size() + 1);
        v->queue.push_back(r.get());
        run->reqs.push_back(r.release());                 // (cannot throw: reserved)
        ++run->pending;
      }
    }
  } catch (const std::bad_alloc&) {
    if (run) {                                            // what was queued of this block leaves the queue again
      for (zkgpu_request* r : run->reqs) {
        for (auto it = v->queue.begin(); it != v->queue.end(); ++it) if (*it == r) { v->queue.erase(it); break; }
        delete r;
      }
      v->block_runs.erase(run->id);
    }
    v->last_error = "out of host memory while queueing the block's batches";
    return nullptr;
  }
  (void)ticket_dispatch(v, now);
  return run;
}

// has the device finished every batch of the run?  (never blocks; v->mu held)
bool block_done(zkgpu_verifier* v, zkgpu_verifier::BlockRun* run) {
  DeviceGuard g(v->root->device);
  for (zkgpu_request* r : run->reqs) {
    if (r->state == 0) return false;
    if (r->state == 1 && hipEventQuery(v->lanes[(size_t)r->lane]->ev_done) == hipErrorNotReady) return false;
  }
  return true;
}

// waits for the block's batches (always all of them: nothing of the run is left queued or on a lane after an error); v->mu held
int block_finish(zkgpu_verifier* v, zkgpu_verifier::BlockRun* run, uint8_t* accept_bitmap) {
  while (run->pending) {
    zkgpu_request* r = run->reqs.back();                // (collected requests leave the list: see ticket_collect)
    if (r->state == 0) (void)ticket_dispatch(v, true);  // everything that is queued goes out, merged by shape across blocks
    else if (r->state == 1) ticket_collect(v, r->lane);
  }
  const int rc = run->rc;
  const size_t nbytes = (run->b->batch + 7) / 8;
  if (rc == ZKGPU_OK) memcpy(accept_bitmap, run->bits.data(), nbytes); else memset(accept_bitmap, 0, nbytes);
  v->block_runs.erase(run->id);
  return rc;
}

// a block's request has its verdicts (or its error): into the run, and the request is gone (v->mu held)
void block_request_done(zkgpu_verifier* v, zkgpu_request* r) {
  zkgpu_verifier::BlockRun* run = (zkgpu_verifier::BlockRun*)r->run;
  if (r->rc == ZKGPU_OK) {
    const auto& idx = run->b->groups[r->group].idx;
    for (size_t j = 0; j < r->batch; ++j)
      if ((r->bits[j / 8] >> (j % 8)) & 1) { const uint32_t i = idx[r->off + j]; run->bits[i / 8] |= (uint8_t)(1u << (i % 8)); }
  } else if (run->rc == ZKGPU_OK) {
    run->rc = r->rc;
  }
  for (auto it = run->reqs.begin(); it != run->reqs.end(); ++it) if (*it == r) { run->reqs.erase(it); break; }
  --run->pending;
  delete r;
}

// the oldest device batch in flight is waited for (v->mu held; there is one)
void collect_oldest(zkgpu_verifier* v) { ticket_collect(v, v->busy.front()); }

}  // namespace

int zkgpu_verifier_verify_block(zkgpu_verifier* v, const zkgpu_txblock* b, uint8_t* accept_bitmap) {
  if (!v || !b || b->v != v || !accept_bitmap) return ZKGPU_EINVAL;
  memset(accept_bitmap, 0, (b->batch + 7) / 8);
  std::lock_guard<std::mutex> lk(v->mu);
  zkgpu_verifier::BlockRun* run = block_start(v, b, true);
  return run ? block_finish(v, run, accept_bitmap) : ZKGPU_ENOMEM;
}

// The same in two halves, so that the next block's batches are on the lanes before the last one's verdicts are waited
// for (a node verifying a stream of blocks).  The block must stay alive until its run has been finished; a run that is
// never finished is drained when the verifier is destroyed.
int zkgpu_verifier_block_start(zkgpu_verifier* v, const zkgpu_txblock* b, uint64_t* run_id) {
  if (!v || !b || b->v != v || !run_id) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  zkgpu_verifier::BlockRun* run = block_start(v, b, false);   // (launched when the merge target is reached, or when a run is finished)
  if (!run) return ZKGPU_ENOMEM;
  *run_id = run->id;
  return ZKGPU_OK;
}

int zkgpu_verifier_block_finish(zkgpu_verifier* v, uint64_t run_id, uint8_t* accept_bitmap) {
  if (!v || !accept_bitmap) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  auto it = v->block_runs.find(run_id);
  if (it == v->block_runs.end()) return ZKGPU_EINVAL;
  return block_finish(v, it->second.get(), accept_bitmap);
}

// Host-memory form: block -> HBM -> verdicts (PCIe copies included).
int zkgpu_verifier_verify(zkgpu_verifier* v, size_t batch, const uint32_t* n_in, const uint32_t* n_out,
                          const uint8_t* commitments, const uint8_t* proofs, const uint64_t* proof_offsets,
                          const uint8_t* r_bytes, uint8_t* accept_bitmap) {
  if (!v || !accept_bitmap) return ZKGPU_EINVAL;
  memset(accept_bitmap, 0, (batch + 7) / 8);
  if (batch == 0) return ZKGPU_OK;
  zkgpu_txblock* b = nullptr;
  TRY(zkgpu_txblock_create(v, batch, n_in, n_out, commitments, proofs, proof_offsets, r_bytes, &b));
  const int rc = zkgpu_verifier_verify_block(v, b, accept_bitmap);
  zkgpu_txblock_destroy(b);
  return rc;
}

// ---- tickets: many small batches in flight, merged into few large device batches ----------------------
// A 1024-transaction batch is ONE round of workgroups for every chip-filling kernel: each pays its ramp and its
// tail, and neighbours cannot share a CU whose register file one of them fills; a 4096-transaction batch runs
// ~1.5x faster per transaction.  zkgpu_verifier_submit_dev therefore only QUEUES a batch (inputs resident in
// HBM, uniform shape) and returns a ticket; queued batches of one shape are merged, up to merge_target
// transactions, copied side by side into a lane's workspace (device to device, ~1.4 KB per transaction) and
// launched as one batch as soon as the target is reached and a lane is free -- or when someone waits for one of
// them.  zkgpu_verifier_wait returns that batch's own accept bitmap.  Verdicts are those of separate batches;
// what changes is when a batch starts.  (A runtime policy, as dynamic batching in a serving system; the reference
// has no counterpart.)  The inputs must stay valid until the ticket has been waited for.
namespace {

void ticket_collect(zkgpu_verifier* v, int lane) {       // v->mu held
  std::vector<zkgpu_request*> members;
  members.swap(v->running[(size_t)lane]);
  for (auto it = v->busy.begin(); it != v->busy.end(); ++it) if (*it == lane) { v->busy.erase(it); break; }
  if (members.empty()) return;
  size_t total = 0;
  for (auto* r : members) total += r->batch;
  std::vector<uint8_t> big((total + 7) / 8, 0);
  const int rc = zkgpu_verify_wait(v->lanes[(size_t)lane], big.data());
  if (rc != ZKGPU_OK) v->last_error = zkgpu_last_error(v->lanes[(size_t)lane]);
  if ((size_t)lane < v->lane_stage.size() && v->lane_stage[(size_t)lane] >= 0) {      // a batch formed from host memory: its twin is free
    v->host_stages[(size_t)v->lane_stage[(size_t)lane]].taken = false;
    v->lane_stage[(size_t)lane] = -1;
  }
  for (auto* r : members) {
    r->bits.assign((r->batch + 7) / 8, 0);
    if (rc == ZKGPU_OK)
      for (size_t i = 0; i < r->batch; ++i) {
        const size_t b = r->bit_off + i;
        if ((big[b / 8] >> (b % 8)) & 1) r->bits[i / 8] |= (uint8_t)(1u << (i % 8));
      }
    r->rc = rc;
    r->state = 2;
    if (r->run) block_request_done(v, r);                // (a block's batch: its verdicts go to the block's run)
  }
}

// launches the batches at the head of the queue that share its shape; force: even below the merge target, and
// if no lane is free the oldest one in flight is collected first
int ticket_dispatch(zkgpu_verifier* v, bool force) {       // v->mu held
  // Device batches that leave in one call are queued piece by piece (zkgpu_ctx::enqueue_phase): every batch's light front, then
  // every batch's point decoding, then the rest of each -- the second batch's transcript and decoding then run beside the
  // first's.  `fronts`: batches whose later pieces are still owed.
  struct Front { int lane; zkgpu_cloak_plan* plan; size_t total, proof_len; const void *com, *proofs, *r; };
  std::vector<Front> fronts;
  auto flush_backs = [&]() {
    std::vector<int> mid_rc(fronts.size(), ZKGPU_OK);
    for (size_t i = 0; i < fronts.size(); ++i) {        // every batch's chip-filling decoding, then every batch's rest
      const Front& f = fronts[i];
      zkgpu_ctx* L = v->lanes[(size_t)f.lane];
      { std::lock_guard<std::recursive_mutex> lk(L->mu); L->enqueue_phase = zkgpu_ctx::ENQ_MID; }
      mid_rc[i] = zkgpu_cloak_verify_submit_dev(L, v->ps, f.plan, f.total, f.com, f.proofs, f.proof_len, f.r);
      { std::lock_guard<std::recursive_mutex> lk(L->mu); L->enqueue_phase = zkgpu_ctx::ENQ_ALL; }
    }
    for (size_t i = 0; i < fronts.size(); ++i) {
      const Front& f = fronts[i];
      zkgpu_ctx* L = v->lanes[(size_t)f.lane];
      int rc = mid_rc[i];
      if (rc == ZKGPU_OK) {
        { std::lock_guard<std::recursive_mutex> lk(L->mu); L->enqueue_phase = zkgpu_ctx::ENQ_BACK; }
        rc = zkgpu_cloak_verify_submit_dev(L, v->ps, f.plan, f.total, f.com, f.proofs, f.proof_len, f.r);
      }
      { std::lock_guard<std::recursive_mutex> lk(L->mu); L->enqueue_phase = zkgpu_ctx::ENQ_ALL; L->awaiting_back = false; }
      if (rc != ZKGPU_OK) {
        // the back half could not be queued (a launch failed): what the front half queued is waited for, and the batch's
        // requests fail with the error -- nothing is left in flight on the lane
        v->last_error = zkgpu_last_error(L);
        quiesce_after_fault(L);
        std::vector<zkgpu_request*> members;
        members.swap(v->running[(size_t)f.lane]);
        for (auto it = v->busy.begin(); it != v->busy.end(); ++it) if (*it == f.lane) { v->busy.erase(it); break; }
        for (zkgpu_request* r : members) {
          r->state = 2; r->rc = rc; r->bits.assign((r->batch + 7) / 8, 0);
          if (r->run) block_request_done(v, r);
        }
      }
    }
    fronts.clear();
  };
  struct FlushAtExit { decltype(flush_backs)& f; ~FlushAtExit() { f(); } } flush_at_exit{flush_backs};
  size_t owed = 0;                                      // device batches still to leave of the cut decided for the head's shape
  uint64_t owed_shape[3] = {0, 0, 0};
  while (!v->queue.empty()) {
    zkgpu_request* head = v->queue.front();
    // Everything queued of the head's shape, T transactions, leaves in round(T / target) device batches of EQUAL size (to the
    // ticket): 20 tickets of 1024 at a target of 10 240 are two batches of ten as before, but 16 are two of eight (not ten and
    // six), 24 two of twelve, 37 four of ten, nine, nine, nine.  What a run of any length then pays is the one tail behind its
    // last batch, not a short straggler batch as well (`value` used to peak where the step count was a
    // multiple of the target).  Tickets that trickle in one by one leave at the target, as they always did.
    const size_t target = head->run ? v->block_merge : v->merge_target;
    size_t queued = 0;
    for (zkgpu_request* r : v->queue)
      if (r->n_in == head->n_in && r->n_out == head->n_out && r->proof_len == head->proof_len) queued += r->batch;
    // the cut is decided ONCE for what is queued, and every part of it leaves in this call (`owed`): two parts of 8192 must not
    // turn into one now and one when somebody waits (the second would start after the first had finished -- measured, 16 steps)
    const bool same_cut = owed > 0 && owed_shape[0] == head->n_in && owed_shape[1] == head->n_out && owed_shape[2] == head->proof_len;
    if (!same_cut) {
      if (queued < target && !force) return ZKGPU_OK;
      owed = zk::ticket_parts(queued, target);            // (ticket_cut.hpp: the policy, also driven by the CPU tests)
      owed_shape[0] = head->n_in; owed_shape[1] = head->n_out; owed_shape[2] = head->proof_len;
    }
    const size_t quota = zk::ticket_quota(queued, owed);
    std::vector<zkgpu_request*> pick;
    size_t total = 0;
    for (zkgpu_request* r : v->queue) {
      if (r->n_in != head->n_in || r->n_out != head->n_out || r->proof_len != head->proof_len) continue;
      pick.push_back(r);
      total += r->batch;
      if (total >= quota) break;
    }
    int lane = -1;
    for (size_t i = 0; i < v->lanes.size(); ++i) if (v->running[i].empty()) { lane = (int)i; break; }
    if (lane < 0) {
      if (!force) return ZKGPU_OK;
      flush_backs();                                    // (a lane that is waited for must have its whole batch queued)
      collect_oldest(v);
      continue;
    }
    --owed;
    zkgpu_ctx* L = v->lanes[(size_t)lane];
    int rc = ZKGPU_OK;
    std::string plan_err;
    zkgpu_cloak_plan* plan = verifier_plan(v, head->n_in, head->n_out, &rc, &plan_err);   // rc != OK: no plan THIS time -> the tickets fail with it
    if (rc != ZKGPU_OK) v->last_error = plan_err;                                          // (v->mu held)
    const void *p_com = pick[0]->d_com, *p_proofs = pick[0]->d_proofs, *p_r = pick[0]->d_r;
    // batches of blocks whose copy to HBM is still queued on the copy stream: whoever reads their inputs waits for it
    if (pick.size() == 1) {
      if (pick[0]->ready) { std::lock_guard<std::recursive_mutex> lk(L->mu); L->dep_event = pick[0]->ready; }
    } else {
      std::lock_guard<std::recursive_mutex> lk(L->mu);
      DeviceGuard g(L->device);
      hipEvent_t seen = nullptr;
      for (zkgpu_request* r : pick)
        if (r->ready && r->ready != seen) { seen = r->ready; if (hipStreamWaitEvent(L->stream_l, r->ready, 0) != hipSuccess && rc == ZKGPU_OK) { L->last_error = "hipStreamWaitEvent (block copy)"; rc = ZKGPU_EHIP; } }
    }
    if (rc == ZKGPU_OK && plan && proof_len_fits(plan->shape, head->proof_len) && pick.size() > 1) {
      std::lock_guard<std::recursive_mutex> lk(L->mu);
      DeviceGuard g(L->device);
      const size_t wcom = (size_t)plan->shape.m * 32;
      rc = ensure(L, L->coal_com, total * wcom);
      if (rc == ZKGPU_OK) rc = ensure(L, L->coal_proofs, total * head->proof_len + 16);   // (k_merge_inputs writes whole words)
      if (rc == ZKGPU_OK) rc = ensure(L, L->coal_r, total * 64);
      // one gather launch per 16 queued batches (commitments and randomness as 16-byte vectors; buffers that are not
      // 16-byte aligned go through plain copies)
      size_t off = 0, at = 0;
      while (rc == ZKGPU_OK && at < pick.size()) {
        MergeSources ms;
        ms.n = 0;
        bool aligned = true;
        const size_t base = off;
        while (at < pick.size() && ms.n < 16) {
          zkgpu_request* r = pick[at];
          aligned = aligned && ((uintptr_t)r->d_com % 16 == 0) && ((uintptr_t)r->d_r % 16 == 0);
          ms.com[ms.n] = (const uint4*)r->d_com; ms.proofs[ms.n] = (const uint8_t*)r->d_proofs; ms.r[ms.n] = (const uint4*)r->d_r;
          ms.first[ms.n] = (uint32_t)(off - base);
          off += r->batch;
          ++ms.n; ++at;
        }
        ms.first[ms.n] = (uint32_t)(off - base);
        if (aligned && (base * head->proof_len) % 4 == 0) {
          const uint32_t span = ms.first[ms.n];
          hipLaunchKernelGGL(k_merge_inputs, dim3(std::min<unsigned>(blocks_for((uint64_t)span * (head->proof_len / 4 + 1), 256), 1024u)), dim3(256), 0, L->stream_l, ms,
                             (uint32_t)(wcom / 16), (uint32_t)head->proof_len, (uint4*)((char*)L->coal_com.p + base * wcom),
                             (uint8_t*)L->coal_proofs.p + base * head->proof_len, (uint4*)((char*)L->coal_r.p + base * 64));
          if (hipGetLastError() != hipSuccess) { L->last_error = "k_merge_inputs"; rc = ZKGPU_EHIP; }
        } else {
          size_t o2 = base;
          for (uint32_t k = 0; k < ms.n && rc == ZKGPU_OK; ++k) {
            const size_t nb = ms.first[k + 1] - ms.first[k];
            hipError_t e = hipMemcpyAsync((char*)L->coal_com.p + o2 * wcom, ms.com[k], nb * wcom, hipMemcpyDeviceToDevice, L->stream_l);
            if (e == hipSuccess) e = hipMemcpyAsync((char*)L->coal_proofs.p + o2 * head->proof_len, ms.proofs[k], nb * head->proof_len, hipMemcpyDeviceToDevice, L->stream_l);
            if (e == hipSuccess) e = hipMemcpyAsync((char*)L->coal_r.p + o2 * 64, ms.r[k], nb * 64, hipMemcpyDeviceToDevice, L->stream_l);
            if (e != hipSuccess) { L->last_error = hipGetErrorString(e); rc = ZKGPU_EHIP; }
            o2 += nb;
          }
        }
      }
      p_com = L->coal_com.p; p_proofs = L->coal_proofs.p; p_r = L->coal_r.p;
    }
    bool front_only = false;
    if (rc == ZKGPU_OK && plan) {
      { std::lock_guard<std::recursive_mutex> lk(L->mu); L->enqueue_phase = zkgpu_ctx::ENQ_FRONT; L->awaiting_back = false; }
      rc = zkgpu_cloak_verify_submit_dev(L, v->ps, plan, total, p_com, p_proofs, head->proof_len, p_r);
      { std::lock_guard<std::recursive_mutex> lk(L->mu); L->enqueue_phase = zkgpu_ctx::ENQ_ALL; front_only = rc == ZKGPU_OK && L->awaiting_back; }
      if (front_only) fronts.push_back(Front{lane, plan, total, head->proof_len, p_com, p_proofs, p_r});
    }
    size_t off = 0;
    for (zkgpu_request* r : pick) {
      for (auto it = v->queue.begin(); it != v->queue.end(); ++it) if (*it == r) { v->queue.erase(it); break; }
      r->bit_off = off; off += r->batch;
      if (rc == ZKGPU_OK && plan) { r->state = 1; r->lane = lane; }
      else { r->state = 2; r->rc = rc; r->bits.assign((r->batch + 7) / 8, 0); }   // rc OK and no plan: every proof is Err
    }
    if (rc == ZKGPU_OK && plan) { v->running[(size_t)lane] = pick; v->busy.push_back(lane); }
    else {
      if (rc != ZKGPU_OK && plan) v->last_error = zkgpu_last_error(L);
      if (rc != ZKGPU_OK) quiesce_after_fault(L);          // (merge copies, a front half: nothing of it stays in flight on a lane that reads as free)
      { std::lock_guard<std::recursive_mutex> lk(L->mu); L->dep_event = nullptr; }
      for (zkgpu_request* r : pick) if (r->run) block_request_done(v, r);     // (a block's batches answer to their run, at once)
    }
  }
  return ZKGPU_OK;
}

}  // namespace

namespace {

// ---- host-memory tickets (v->mu held throughout) --------------------------------------------------------------------
int free_ticket_lane(zkgpu_verifier* v) {
  for (size_t i = 0; i < v->lanes.size(); ++i) if (v->running[i].empty()) return (int)i;
  return -1;
}

// a staging area nobody forms a batch in; its last copy to the device is waited for (long done in practice).  -1: none
int host_stage_acquire(zkgpu_verifier* v, size_t bytes, int* rc) {
  *rc = ZKGPU_OK;
  if (v->host_stages.empty()) v->host_stages.resize(2 * v->lanes.size() + 2);     // (one per batch in flight, as many being formed, two spare)
  // Every FREE area is brought to the size asked for, not only the one handed out: areas stay taken while their batch is in
  // flight, so which area a batch gets depends on timing, and an area met for the first time in the middle of a burst would
  // cost a hipHostMalloc + hipMalloc there (both synchronise the device: one bench run in three lost half its rate to it).
  for (size_t i = 0; i < v->host_stages.size(); ++i) {
    zkgpu_verifier::HostStage& hs = v->host_stages[i];
    if (hs.taken || hs.cap >= bytes) continue;
    DeviceGuard g(v->root->device);
    if (hs.pin) (void)hipHostFree(hs.pin);
    if (hs.dev) (void)hipFree(hs.dev);
    hs.pin = hs.dev = nullptr; hs.cap = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    if (hipHostMalloc(&hs.pin, want, hipHostMallocDefault) != hipSuccess) { hs.pin = nullptr; continue; }     // (tried again when it is handed out)
    if (hipMalloc(&hs.dev, want) != hipSuccess) { (void)hipHostFree(hs.pin); hs.pin = hs.dev = nullptr; continue; }
    hs.cap = want;
  }
  for (size_t i = 0; i < v->host_stages.size(); ++i) {
    zkgpu_verifier::HostStage& hs = v->host_stages[i];
    if (hs.taken) continue;                // (a free area has no copy in flight: its last batch has been collected)
    DeviceGuard g(v->root->device);
    if (hs.cap < bytes) {
      if (hs.pin) (void)hipHostFree(hs.pin);
      if (hs.dev) (void)hipFree(hs.dev);
      hs.pin = hs.dev = nullptr; hs.cap = 0;
      const size_t want = bytes + bytes / 8 + 4096;
      if (hipHostMalloc(&hs.pin, want, hipHostMallocDefault) != hipSuccess) { hs.pin = nullptr; v->last_error = "hipHostMalloc (ticket staging)"; *rc = ZKGPU_ENOMEM; return -1; }
      if (hipMalloc(&hs.dev, want) != hipSuccess) { (void)hipHostFree(hs.pin); hs.pin = hs.dev = nullptr; v->last_error = "hipMalloc (ticket staging twin)"; *rc = ZKGPU_ENOMEM; return -1; }
      hs.cap = want;
    }
    if (!hs.copied && hipEventCreateWithFlags(&hs.copied, hipEventDisableTiming) != hipSuccess) { v->last_error = "hipEventCreate (ticket staging)"; *rc = ZKGPU_EHIP; return -1; }
    if (!v->copy_stream && hipStreamCreateWithFlags(&v->copy_stream, hipStreamNonBlocking) != hipSuccess) { v->last_error = "hipStreamCreate (ticket staging)"; *rc = ZKGPU_EHIP; return -1; }
    hs.taken = true;
    return (int)i;
  }
  return -1;
}

void host_batch_fail(zkgpu_verifier* v, zkgpu_host_batch* F, int rc) {      // every member done, with rc (OK: all bits zero)
  for (zkgpu_request* r : F->members) { r->state = 2; r->rc = rc; r->form = nullptr; r->bits.assign((r->batch + 7) / 8, 0); }
  if (F->stage >= 0) {
    zkgpu_verifier::HostStage& hs = v->host_stages[(size_t)F->stage];
    quiesce_after_fault(v->root);      // (copies of staged tickets may still be queued, kernels of a half-queued batch may still read the twin)
    hs.taken = false;
  }
}

// the formed batch F -> lane: its bytes are already on their way to the staging area's twin in HBM (host_submit_one); the lane
// waits for the last of those copies and reads the twin in place
int host_launch(zkgpu_verifier* v, zkgpu_host_batch* F, int lane) {
  zkgpu_ctx* L = v->lanes[(size_t)lane];
  zkgpu_verifier::HostStage& hs = v->host_stages[(size_t)F->stage];
  const char* d = (const char*)hs.dev;
  { std::lock_guard<std::recursive_mutex> lk(L->mu); L->dep_event = hs.copied; }
  const int rc = zkgpu_cloak_verify_submit_dev(L, v->ps, F->plan, F->total, d, d + F->o_proofs, F->proof_len, d + F->o_r);
  if (rc != ZKGPU_OK) {
    v->last_error = zkgpu_last_error(L);
    { std::lock_guard<std::recursive_mutex> lk(L->mu); L->dep_event = nullptr; }
    host_batch_fail(v, F, rc);
    return rc;
  }
  size_t off = 0;
  for (zkgpu_request* r : F->members) { r->bit_off = off; off += r->batch; r->state = 1; r->lane = lane; r->form = nullptr; }
  v->running[(size_t)lane] = F->members;
  v->busy.push_back(lane);
  if (v->lane_stage.size() < v->lanes.size()) v->lane_stage.resize(v->lanes.size(), -1);
  v->lane_stage[(size_t)lane] = F->stage;         // (the area is free again when this lane's batch has been collected)
  return ZKGPU_OK;
}

// launches the formed batches that are full, oldest first, while lanes are free; `must`: that one whatever it holds, and
// if no lane is free the oldest batch in flight is collected first -- also on behalf of OLDER full batches that stand
// before `must` in the list (they keep their order and go out first; returning there instead left `must` where it was
// for ever: a caller that waited for its newest ticket first, with more full batches formed than lanes, never came back)
int host_dispatch(zkgpu_verifier* v, zkgpu_host_batch* must) {
  for (size_t i = 0; i < v->forming.size();) {
    zkgpu_host_batch* F = v->forming[i].get();
    const bool due = F == must || F->total >= std::min(F->cap_tx, v->merge_target);
    if (!due) { ++i; continue; }
    int lane = free_ticket_lane(v);
    if (lane < 0) {
      if (!must) return ZKGPU_OK;        // (full batches keep their order: the next free lane is the oldest one's)
      if (v->busy.empty()) {             // (cannot happen with ≥ 1 lane; never spin on it)
        v->last_error = "host_dispatch: no lane is free and none is busy";
        for (size_t k = 0; k < v->forming.size(); ++k)
          if (v->forming[k].get() == must) { host_batch_fail(v, must, ZKGPU_EINVAL); v->forming.erase(v->forming.begin() + (long)k); break; }
        return ZKGPU_EINVAL;
      }
      collect_oldest(v);
      continue;
    }
    std::unique_ptr<zkgpu_host_batch> owned = std::move(v->forming[i]);
    v->forming.erase(v->forming.begin() + (long)i);
    if (owned.get() == must) must = nullptr;     // (what stands behind it waits for lanes like any other)
    (void)host_launch(v, owned.get(), lane);     // (a failure is recorded in its tickets)
  }
  return ZKGPU_OK;
}

// the caller's bytes -> pinned staging memory: on one thread ~10 GB/s, i.e. 0.14 ms per 1024-transaction ticket and 1.4 ms
// before the FIRST device batch of a burst can leave; pieces of 8 KB on up to eight pool threads beyond 256 KB
// Up to three spans of one ticket in ONE pass over the pool (a pass costs ~30 us to wake the workers whatever it copies: three
// passes per ticket were most of the 0.1 ms a ticket took to stage -- and the first device batch of a burst waits for ten).
struct CopySpan { void* dst; const void* src; size_t n; };
void staged_copy(const CopySpan* spans, int count) {
  constexpr size_t PIECE = 8u << 10;
  size_t total = 0, pieces[3] = {0, 0, 0};
  for (int k = 0; k < count; ++k) { total += spans[k].n; pieces[k] = (spans[k].n + PIECE - 1) / PIECE; }
  if (total < (256u << 10)) { for (int k = 0; k < count; ++k) memcpy(spans[k].dst, spans[k].src, spans[k].n); return; }
  host_parallel(pieces[0] + pieces[1] + pieces[2], std::min(usable_cpus(), 8), [&](size_t i) {
    int k = 0;
    while (i >= pieces[k]) { i -= pieces[k]; ++k; }
    const size_t at = i * PIECE;
    memcpy((char*)spans[k].dst + at, (const char*)spans[k].src + at, std::min(PIECE, spans[k].n - at));
  });
}

// one ticket from host memory: copied into the batch being formed for its shape (a new one when that is full)
int host_submit_one(zkgpu_verifier* v, uint32_t n_in, uint32_t n_out, size_t batch, const uint8_t* com, const uint8_t* proofs,
                    size_t proof_len, const uint8_t* r_bytes, uint64_t* ticket) {
  zkgpu_request* r = new zkgpu_request();
  r->id = v->next_id++;
  r->n_in = n_in; r->n_out = n_out; r->batch = batch; r->proof_len = proof_len;
  v->requests[r->id] = r;
  *ticket = r->id;
  int rc = ZKGPU_OK;
  std::string plan_err;
  zkgpu_cloak_plan* plan = verifier_plan(v, n_in, n_out, &rc, &plan_err);
  if (rc != ZKGPU_OK) v->last_error = plan_err;
  if (rc != ZKGPU_OK || !plan || !proof_len_fits(plan->shape, proof_len)) {   // no plan THIS time: the ticket fails with rc; a shape or
    r->state = 2; r->rc = rc; r->bits.assign((batch + 7) / 8, 0);             // length the reference rejects: every proof is Err
    return ZKGPU_OK;
  }
  const size_t wcom = (size_t)plan->shape.m * 32;
  zkgpu_host_batch* F = nullptr;
  for (auto& f : v->forming)
    if (f->n_in == n_in && f->n_out == n_out && f->proof_len == proof_len && f->total < std::min(f->cap_tx, v->merge_target) && f->total + batch <= f->cap_tx) F = f.get();
  while (!F) {
    const size_t cap_tx = std::max(v->merge_target, batch);
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_proofs = align(cap_tx * wcom), o_r = align(o_proofs + cap_tx * proof_len + 16), bytes = o_r + cap_tx * 64;
    const int st = host_stage_acquire(v, bytes, &rc);
    if (rc != ZKGPU_OK) { r->state = 2; r->rc = rc; r->bits.assign((batch + 7) / 8, 0); return ZKGPU_OK; }
    if (st < 0) {                         // every area holds a batch that waits: the oldest goes out now (a lane is made free for it)
      (void)host_dispatch(v, v->forming.front().get());
      continue;
    }
    std::unique_ptr<zkgpu_host_batch> nf(new zkgpu_host_batch());
    nf->n_in = n_in; nf->n_out = n_out; nf->proof_len = proof_len; nf->wcom = wcom; nf->cap_tx = cap_tx;
    nf->o_proofs = o_proofs; nf->o_r = o_r; nf->stage = st; nf->plan = plan;
    F = nf.get();
    v->forming.push_back(std::move(nf));
  }
  zkgpu_verifier::HostStage& hs = v->host_stages[(size_t)F->stage];
  char* h = (char*)hs.pin;
  {
    const CopySpan spans[3] = {{h + F->total * wcom, com, batch * wcom}, {h + F->o_proofs + F->total * proof_len, proofs, batch * proof_len},
                               {h + F->o_r + F->total * 64, r_bytes, batch * 64}};
    staged_copy(spans, r_bytes ? 3 : 2);
  }
  if (!r_bytes) {                                // verifier randomness: SHAKE256 of 32 bytes from the OS and the ticket number
    uint8_t seed[40] = {0};
    if (!os_random(seed, 32)) { r->state = 2; r->rc = ZKGPU_EINVAL; r->bits.assign((batch + 7) / 8, 0); v->last_error = "getrandom failed"; return ZKGPU_OK; }
    for (int q = 0; q < 8; ++q) seed[32 + q] = (uint8_t)(r->id >> (8 * q));
    Sponge sp = shake256_sponge();
    sp.absorb(seed, 40);
    sp.squeeze((uint8_t*)h + F->o_r + F->total * 64, batch * 64);
  }
  {
    // this ticket's three pieces -> the twin, now: the copy runs beside the staging of the batch's later tickets
    DeviceGuard g(v->root->device);
    char* d = (char*)hs.dev;
    const size_t a = F->total * wcom, b = F->o_proofs + F->total * proof_len, c = F->o_r + F->total * 64;
    hipError_t e = hipMemcpyAsync(d + a, h + a, batch * wcom, hipMemcpyHostToDevice, v->copy_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + b, h + b, batch * proof_len, hipMemcpyHostToDevice, v->copy_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + c, h + c, batch * 64, hipMemcpyHostToDevice, v->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(hs.copied, v->copy_stream);
    if (e != hipSuccess) {
      v->last_error = std::string("ticket staging copy: ") + hipGetErrorString(e);
      quiesce_after_fault(v->root);
      r->state = 2; r->rc = ZKGPU_EHIP; r->bits.assign((batch + 7) / 8, 0);
      return ZKGPU_OK;
    }
  }
  r->form = F;
  F->members.push_back(r);
  F->total += batch;
  return ZKGPU_OK;
}

}  // namespace

// Tickets from HOST memory: as zkgpu_verifier_submit_dev, but the inputs lie in the caller's memory and are free again
// when the call returns (they are copied into pinned staging memory here).  r_bytes == NULL: verifier randomness from the OS.
int zkgpu_verifier_submit(zkgpu_verifier* v, uint32_t n_in, uint32_t n_out, size_t batch, const uint8_t* commitments,
                          const uint8_t* proofs, size_t proof_len, const uint8_t* r_bytes, uint64_t* ticket) {
  if (!v || !ticket || batch == 0 || batch >= (1ull << 24) || !commitments || !proofs) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  TRY(host_submit_one(v, n_in, n_out, batch, commitments, proofs, proof_len, r_bytes, ticket));
  return host_dispatch(v, nullptr);
}

int zkgpu_verifier_submit_many(zkgpu_verifier* v, uint32_t n_in, uint32_t n_out, size_t count, size_t batch_each,
                               const uint8_t* const* commitments, const uint8_t* const* proofs, size_t proof_len,
                               const uint8_t* const* r_bytes, uint64_t* tickets) {
  if (!v || !tickets || !commitments || !proofs || batch_each == 0 || batch_each >= (1ull << 24)) return ZKGPU_EINVAL;
  for (size_t i = 0; i < count; ++i) if (!commitments[i] || !proofs[i]) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  for (size_t i = 0; i < count; ++i) {
    TRY(host_submit_one(v, n_in, n_out, batch_each, commitments[i], proofs[i], proof_len, r_bytes ? r_bytes[i] : nullptr, &tickets[i]));
    (void)host_dispatch(v, nullptr);      // a batch that became full goes out while the next is being formed
  }
  return ZKGPU_OK;
}

// Sizes every lane's workspace (and its merge buffers) for device batches of `transactions` statements of the shape, so that no
// lane allocates when it first meets a batch that large: hipMalloc / hipFree synchronise the device, and a verifier whose
// batches vary in shape and size (blocks of mixed shapes merged across blocks in flight) otherwise keeps growing workspaces for
// many calls (measured, config 4 with three blocks in flight: steps of 12 - 20 ms among steps of 3 ms).  A shape the
// generator set cannot serve: ZKGPU_OK, nothing to do.
int zkgpu_verifier_reserve(zkgpu_verifier* v, uint32_t n_in, uint32_t n_out, size_t transactions) {
  if (!v || transactions == 0 || transactions >= (1ull << 24)) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  int rc = ZKGPU_OK;
  std::string err;
  zkgpu_cloak_plan* plan = verifier_plan(v, n_in, n_out, &rc, &err);
  if (rc != ZKGPU_OK) { v->last_error = err; return rc; }
  if (!plan) return ZKGPU_OK;
  while (!v->busy.empty()) ticket_collect(v, v->busy.front());       // (the lanes must be idle: their workspaces are about to move)
  const size_t proof_len = 1 + 4 * (size_t)plan->shape.proof_words;
  const size_t wcom = (size_t)plan->shape.m * 32;
  for (zkgpu_ctx* L : v->lanes) {
    rc = cloak_reserve(L, v->ps, plan, transactions, proof_len);
    if (rc == ZKGPU_OK) {
      std::lock_guard<std::recursive_mutex> llk(L->mu);
      DeviceGuard g(L->device);
      rc = ensure(L, L->coal_com, transactions * wcom);
      if (rc == ZKGPU_OK) rc = ensure(L, L->coal_proofs, transactions * proof_len + 16);
      if (rc == ZKGPU_OK) rc = ensure(L, L->coal_r, transactions * 64);
    }
    if (rc != ZKGPU_OK) { v->last_error = zkgpu_last_error(L); return rc; }
  }
  return ZKGPU_OK;
}

int zkgpu_verifier_set_merge(zkgpu_verifier* v, size_t transactions) {
  if (!v || transactions == 0 || transactions >= (1u << 24)) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  v->merge_target = v->block_merge = transactions;        // (one knob for a caller who turns it: tickets and blocks alike)
  return ZKGPU_OK;
}

int zkgpu_verifier_submit_dev(zkgpu_verifier* v, uint32_t n_in, uint32_t n_out, size_t batch, const void* d_commitments,
                              const void* d_proofs, size_t proof_len, const void* d_r, uint64_t* ticket) {
  if (!v || !ticket || batch == 0 || batch >= (1ull << 24) || !d_commitments || !d_proofs || !d_r) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  zkgpu_request* r = new zkgpu_request();
  r->id = v->next_id++;
  r->n_in = n_in; r->n_out = n_out; r->batch = batch; r->proof_len = proof_len;
  r->d_com = d_commitments; r->d_proofs = d_proofs; r->d_r = d_r;
  v->requests[r->id] = r;
  v->queue.push_back(r);
  *ticket = r->id;
  return ticket_dispatch(v, false);
}

// `count` batches of one shape and size in one call (what a caller holding a queue of them would do): tickets[count]
int zkgpu_verifier_submit_many_dev(zkgpu_verifier* v, uint32_t n_in, uint32_t n_out, size_t count, size_t batch_each,
                                   const void* const* d_commitments, const void* const* d_proofs, size_t proof_len,
                                   const void* const* d_r, uint64_t* tickets) {
  if (!v || !tickets || !d_commitments || !d_proofs || !d_r || batch_each == 0 || batch_each >= (1ull << 24)) return ZKGPU_EINVAL;
  for (size_t i = 0; i < count; ++i) if (!d_commitments[i] || !d_proofs[i] || !d_r[i]) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  for (size_t i = 0; i < count; ++i) {
    zkgpu_request* r = new zkgpu_request();
    r->id = v->next_id++;
    r->n_in = n_in; r->n_out = n_out; r->batch = batch_each; r->proof_len = proof_len;
    r->d_com = d_commitments[i]; r->d_proofs = d_proofs[i]; r->d_r = d_r[i];
    v->requests[r->id] = r;
    v->queue.push_back(r);
    tickets[i] = r->id;
  }
  return ticket_dispatch(v, false);
}

int zkgpu_verifier_wait(zkgpu_verifier* v, uint64_t ticket, uint8_t* accept_bitmap) {
  if (!v || !accept_bitmap) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  auto it = v->requests.find(ticket);
  if (it == v->requests.end()) return ZKGPU_EINVAL;
  zkgpu_request* r = it->second;
  while (r->state != 2) {
    if (r->state == 0 && r->form) (void)host_dispatch(v, r->form);     // (a host-memory ticket: its batch goes out as it is)
    else if (r->state == 0) (void)ticket_dispatch(v, true);
    else ticket_collect(v, r->lane);
  }
  const int rc = r->rc;
  memset(accept_bitmap, 0, (r->batch + 7) / 8);
  if (rc == ZKGPU_OK) memcpy(accept_bitmap, r->bits.data(), r->bits.size());
  v->requests.erase(it);
  delete r;
  if (!v->queue.empty()) (void)ticket_dispatch(v, false);
  if (!v->forming.empty()) (void)host_dispatch(v, nullptr);
  return rc;
}

// ---- sharding over the GPUs of a node --------------------------------------------------------
// cuts[0] = 0 <= cuts[1] <= ... <= cuts[world] = batch: rank r verifies transactions
// [cuts[r], cuts[r+1]).  Contiguous, balanced by the number of multiscalar-multiplication terms.
int zkgpu_shard_cuts(size_t batch, const uint32_t* n_in, const uint32_t* n_out, int world, uint64_t* cuts) {
  if (world <= 0 || !cuts || (batch && (!n_in || !n_out))) return ZKGPU_EINVAL;
  std::map<std::pair<uint32_t, uint32_t>, uint64_t> cost;
  std::vector<uint64_t> prefix(batch + 1, 0);
  for (size_t i = 0; i < batch; ++i) {
    const auto key = std::make_pair(n_in[i], n_out[i]);
    auto it = cost.find(key);
    if (it == cost.end()) it = cost.emplace(key, std::max<uint64_t>(1, cloak_msm_terms(n_in[i], n_out[i]))).first;
    prefix[i + 1] = prefix[i] + it->second;
  }
  const uint64_t total = prefix[batch];
  cuts[0] = 0;
  size_t row = 0;
  for (int r = 1; r < world; ++r) {
    // 128-bit product: total * r overflows 64 bits only beyond 2^57 terms
    const uint64_t target = (uint64_t)(((unsigned __int128)total * (unsigned)r) / (unsigned)world);
    while (row < batch && prefix[row + 1] <= target) ++row;
    cuts[r] = row;
  }
  cuts[world] = batch;
  return ZKGPU_OK;
}

// Test hook: world > 0 switches the collective function table to the in-process mock of a world of `world` ranks whose
// OTHER ranks contribute `peer_slots` (world x slot_bytes bytes, rank-major; the place of the rank the communicator is
// then created with is overwritten by what that rank really sends); world = 0 switches back to RCCL.  Returns the number
// of all-gathers the mock has served so far.
long long zkgpu_debug_comm_mock(zkgpu_ctx* ctx, int world, const uint8_t* peer_slots, size_t slot_bytes) {
  if (!test_hooks_enabled()) return ZKGPU_EINVAL;        // (ZKGPU_TEST_HOOKS=1 in the environment when the library was loaded)
  std::lock_guard<std::mutex> lk(g_comm_mock_mu);
  CommMock& m = g_comm_mock;
  if (world < 0) return (long long)m.gathers;            // (a query: nothing changes)
  if (world == 0) { m.on = false; return (long long)m.gathers; }
  zk::fault::Suppress outside_the_product;               // (the mock stands for RCCL and the peers: not what fault injection tests)
  if (!ctx || !peer_slots || slot_bytes == 0 || slot_bytes > COMM_MAX_SLOT) return ZKGPU_EINVAL;
  DeviceGuard g(ctx->device);
  if (m.d_peers) { (void)hipFree(m.d_peers); m.d_peers = nullptr; }
  if (hipMalloc(&m.d_peers, slot_bytes * (size_t)world) != hipSuccess) return ZKGPU_ENOMEM;
  if (hipMemcpy(m.d_peers, peer_slots, slot_bytes * (size_t)world, hipMemcpyHostToDevice) != hipSuccess) return ZKGPU_EHIP;
  m.world = world; m.slot = slot_bytes; m.on = true;
  return (long long)m.gathers;
}

int zkgpu_comm_unique_id(uint8_t id[ZKGPU_COMM_ID_BYTES]) {
  if (!id) return ZKGPU_EINVAL;
  static_assert(sizeof(ncclUniqueId) == ZKGPU_COMM_ID_BYTES, "unique id size");
  RcclApi& api = rccl_for_new_comm();
  if (!api.ok()) return ZKGPU_ENOCOMM;
  ncclUniqueId uid;
  if (api.GetUniqueId(&uid) != ncclSuccess) return ZKGPU_ENOCOMM;
  memcpy(id, &uid, sizeof uid);
  return ZKGPU_OK;
}

int zkgpu_comm_create(zkgpu_ctx* ctx, int rank, int world, const uint8_t id[ZKGPU_COMM_ID_BYTES], zkgpu_comm** out) {
  if (!ctx || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return ZKGPU_EINVAL;
  *out = nullptr;
  std::unique_ptr<zkgpu_comm> cm(new zkgpu_comm());
  cm->ctx = ctx; cm->rank = rank; cm->world = world;
  if (world > 1 || id) {
    RcclApi& api = rccl_for_new_comm();
    if (!api.ok()) { ctx->last_error = api.error; return ZKGPU_ENOCOMM; }
    cm->api = &api;
    DeviceGuard g(ctx->device);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    hipError_t e = hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&cm->d_send, COMM_MAX_SLOT);
    if (e == hipSuccess) e = hipMalloc(&cm->d_recv, COMM_MAX_SLOT * (size_t)world);
    if (e == hipSuccess) e = hipHostMalloc(&cm->h_pin, COMM_MAX_SLOT * ((size_t)world + 1), hipHostMallocDefault);
    if (e == hipSuccess) {
      const uint32_t poison = COMM_POISON;
      e = hipMemcpy(cm->d_send, &poison, 4, hipMemcpyHostToDevice);
    }
    ncclResult_t r = ncclSuccess;
    if (e == hipSuccess) r = api.CommInitRank(&cm->comm, world, uid, rank);
    if (e != hipSuccess || r != ncclSuccess) {
      ctx->last_error = e != hipSuccess ? std::string("zkgpu_comm_create: ") + hipGetErrorString(e)
                                        : std::string("ncclCommInitRank: ") + api.GetErrorString(r);
      if (cm->stream) (void)hipStreamDestroy(cm->stream);
      if (cm->d_send) (void)hipFree(cm->d_send);
      if (cm->d_recv) (void)hipFree(cm->d_recv);
      if (cm->h_pin) (void)hipHostFree(cm->h_pin);
      return e != hipSuccess ? (e == hipErrorOutOfMemory ? ZKGPU_ENOMEM : ZKGPU_EHIP) : ZKGPU_ENOCOMM;
    }
  }
  *out = cm.release();
  return ZKGPU_OK;
}

void zkgpu_comm_destroy(zkgpu_comm* cm) {
  if (!cm) return;
  DeviceGuard g(cm->ctx->device);
  if (cm->comm) (void)comm_api(cm).CommDestroy(cm->comm);
  if (cm->stream) (void)hipStreamDestroy(cm->stream);
  if (cm->d_send) (void)hipFree(cm->d_send);
  if (cm->d_recv) (void)hipFree(cm->d_recv);
  if (cm->h_pin) (void)hipHostFree(cm->h_pin);
  delete cm;
}

int zkgpu_comm_rank(const zkgpu_comm* cm) { return cm ? cm->rank : -1; }
int zkgpu_comm_world(const zkgpu_comm* cm) { return cm ? cm->world : 0; }

// Every rank contributes `bytes` bytes; `all` receives world * bytes (rank-major), in host memory.
// Once the arguments are accepted (they are the same on every rank: `bytes` must be), this rank ENTERS the collective
// whatever happens locally, so that no peer is left waiting in it: the buffers exist since zkgpu_comm_create, and if
// the copy of this rank's contribution to the device fails, what the peers receive from it starts with the poison word
// every call leaves behind in d_send -- a non-zero status to zkgpu_comm_allgather_bitmap.
int zkgpu_comm_allgather(zkgpu_comm* cm, const uint8_t* local, size_t bytes, uint8_t* all) {
  if (!cm || !all || (bytes && !local) || bytes > COMM_MAX_SLOT) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(cm->mu);
  if (bytes == 0) return ZKGPU_OK;
  if (!cm->comm) {                      // a world of one without RCCL
    if (cm->world != 1) return ZKGPU_ENOCOMM;
    memmove(all, local, bytes);
    return ZKGPU_OK;
  }
  zkgpu_ctx* c = cm->ctx;
  DeviceGuard g(c->device);
  const size_t total = bytes * (size_t)cm->world;
  char* h = (char*)cm->h_pin;
  memcpy(h, local, bytes);
  int rc = ZKGPU_OK;
  auto note = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == ZKGPU_OK) { c->last_error = std::string(what) + ": " + hipGetErrorString(e); rc = ZKGPU_EHIP; }
  };
  note(hipMemcpyAsync(cm->d_send, h, bytes, hipMemcpyHostToDevice, cm->stream), "zkgpu_comm_allgather: copy in");
  const ncclResult_t r = comm_api(cm).AllGather(cm->d_send, cm->d_recv, bytes, ncclUint8, cm->comm, cm->stream);
  if (r != ncclSuccess && rc == ZKGPU_OK) { c->last_error = std::string("ncclAllGather: ") + comm_api(cm).GetErrorString(r); rc = ZKGPU_ENOCOMM; }
  note(hipMemcpyAsync(h + COMM_MAX_SLOT, cm->d_recv, total, hipMemcpyDeviceToHost, cm->stream), "zkgpu_comm_allgather: copy out");
  {
    static const uint32_t poison = COMM_POISON;        // (pageable source of an async copy: staged by the runtime)
    note(hipMemcpyAsync(cm->d_send, &poison, 4, hipMemcpyHostToDevice, cm->stream), "zkgpu_comm_allgather: poison");
  }
  note(hipStreamSynchronize(cm->stream), "zkgpu_comm_allgather: synchronize");
  if (rc != ZKGPU_OK) { quiesce_after_fault(c); memset(all, 0, total); return rc; }     // (the pinned buffer is reused by the next call)
  memcpy(all, h + COMM_MAX_SLOT, total);
  return ZKGPU_OK;
}

// The exchange step of the sharded verification: rank r holds the accept bitmap of transactions
// [cuts[r], cuts[r+1]) (bit 0 = transaction cuts[r]) and the status of its own verification; every rank
// receives the bitmap of the whole batch.  If ANY rank reports an error every rank returns an error
// and an all-zero bitmap: one GPU's fault never turns into an accept, and never into a hang.
int zkgpu_comm_allgather_bitmap(zkgpu_comm* cm, const uint64_t* cuts, const uint8_t* local_bitmap, int local_status,
                                uint8_t* whole_bitmap) {
  if (!cm || !cuts || !whole_bitmap) return ZKGPU_EINVAL;
  const int world = cm->world;
  memset(whole_bitmap, 0, (size_t)((cuts[world] + 7) / 8));
  // (the cuts are the same on every rank, and so are the two answers below: no rank goes on into the collective alone)
  const size_t slot = commframe::slot_bytes(cuts, world);
  if (slot == 0 || slot > COMM_MAX_SLOT) return ZKGPU_EINVAL;
  std::vector<uint8_t> mine(slot), all(slot * (size_t)world, 0);
  // a local fault -- the caller's status, or a missing bitmap -- travels THROUGH the gather as this rank's status word:
  // returning here would leave the peers waiting in the collective
  commframe::pack(mine.data(), slot, cuts, cm->rank, local_bitmap, local_status);
  TRY(zkgpu_comm_allgather(cm, mine.data(), slot, all.data()));
  return commframe::unpack(all.data(), slot, cuts, world, cm->rank, whole_bitmap);
}

// Whole batch in host memory on every rank (a block of transactions as every node of the network
// sees it) -> this rank verifies its shard, the verdicts of all shards are gathered over RCCL.
int zkgpu_verifier_verify_sharded(zkgpu_verifier* v, zkgpu_comm* cm, size_t batch, const uint32_t* n_in,
                                  const uint32_t* n_out, const uint8_t* commitments, const uint8_t* proofs,
                                  const uint64_t* proof_offsets, const uint8_t* r_bytes, uint8_t* accept_bitmap) {
  if (!v || !cm || !accept_bitmap) return ZKGPU_EINVAL;
  memset(accept_bitmap, 0, (batch + 7) / 8);
  if (batch == 0) return ZKGPU_OK;
  if (!n_in || !n_out || !commitments || !proofs || !proof_offsets) return ZKGPU_EINVAL;
  std::vector<uint64_t> cuts((size_t)cm->world + 1);
  TRY(zkgpu_shard_cuts(batch, n_in, n_out, cm->world, cuts.data()));
  const size_t lo = (size_t)cuts[cm->rank], hi = (size_t)cuts[cm->rank + 1];
  std::vector<uint8_t> local((hi - lo + 7) / 8 + 1, 0);
  int rc = ZKGPU_OK;
  if (hi > lo) {
    uint64_t com_lo = 0;
    for (size_t i = 0; i < lo; ++i) com_lo += 64ull * ((uint64_t)n_in[i] + n_out[i]);
    std::vector<uint64_t> po(hi - lo + 1);
    for (size_t i = lo; i <= hi; ++i) po[i - lo] = proof_offsets[i] - proof_offsets[lo];
    rc = zkgpu_verifier_verify(v, hi - lo, n_in + lo, n_out + lo, commitments + com_lo, proofs + proof_offsets[lo],
                               po.data(), r_bytes ? r_bytes + 64 * lo : nullptr, local.data());
  }
  return zkgpu_comm_allgather_bitmap(cm, cuts.data(), local.data(), rc, accept_bitmap);
}

// ---- serialized transactions (Tx::verify / Verifier::verify_tx on transaction bytes) ------
// UNPINNED: the wire format, opcodes and labels are a recollection (zkvm_tx.hpp); nothing under
// /root/reference defines them.  The entry point therefore does nothing until the caller names the format it wants
// (zkgpu_verifier_set_tx_format): by default every transaction is reported as "outside the subset" (status 2), which
// is the answer that sends it to the caller's own VM.
// Per transaction on host threads (zkvm_tx.hpp): wire format, the VM of the payment subset, transaction ID, the terms
// of the signature equation; then for the whole batch on the device: the aggregated keys (one small multiscalar
// multiplication each), the signature equations (multiscalar multiplication == identity), and the cloak proofs as a
// block of mixed shapes.  status[i] (optional): 0 accepted, 1 rejected, 2 outside the subset (the caller's own VM must
// decide; the accept bit is 0).  Fail-closed in BOTH outputs: a status of 0 is written only beside an accept bit of 1,
// at the very end; on any error every live transaction reads "rejected".
// The key and signature stages run synchronously on the verifier's root context, which is also lane 0 of its tickets
// and blocks: the whole call holds the verifier's mutex, and whatever is in flight on the lanes is collected first
// (the verdicts stay with their tickets / runs) -- a synchronous call on a context with a batch in flight would
// overwrite that batch's status words and pinned result buffer (and is refused by the context: refuse_if_pending).
int zkgpu_verifier_set_tx_format(zkgpu_verifier* v, int format) {
  if (!v || (format != 0 && format != ZKGPU_TXFORMAT_RECOLLECTED_V1)) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  v->tx_format = format;
  return ZKGPU_OK;
}

namespace {

// host arrays -> the context's input buffers -> kernels and result copy queued (batch_device_enqueue, value mode);
// msm_values_collect finishes.  The arrays must stay alive until then; the context is marked busy meanwhile.
int msm_values_enqueue(zkgpu_ctx* c, const uint8_t* scalars, const uint8_t* points, const uint64_t* offsets, size_t batch) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  const uint64_t n = offsets[batch];
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
  const int rc = batch_device_enqueue(c, job, true);
  if (rc != ZKGPU_OK) { c->split = zkgpu_ctx::SplitOp{}; return rc; }
  c->pending = true; c->pending_batch = batch;
  return ZKGPU_OK;
}

int split_collect(zkgpu_ctx* c, uint8_t* bitmap, uint8_t* values) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  DeviceGuard g(c->device);
  c->pending = false;
  return batch_collect(c, bitmap, values);
}

// has everything queued by the last *_enqueue on this context run?  (never blocks)
bool split_done(zkgpu_ctx* c) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (c->split.kind == 0 || c->split.batch == 0) return true;
  DeviceGuard g(c->device);
  return hipStreamQuery(c->split.stream) != hipErrorNotReady;
}

// the same for rows of dynamic terms + terms on the resident set's tables (the signature equations)
int verify_ps_enqueue(zkgpu_ctx* c, const zkgpu_pointset* ps, size_t batch, const uint8_t* dyn_scalars, const uint8_t* dyn_points,
                      const uint64_t* dyn_offsets, const uint8_t* static_scalars, const uint32_t* static_index, const uint64_t* static_offsets) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  TRY(refuse_if_pending(c));
  DeviceGuard g(c->device);
  const uint64_t nd = dyn_offsets[batch], ns = static_offsets[batch];
  TRY(upload(c, c->in_scalars, dyn_scalars, nd * 32));
  TRY(upload(c, c->in_points, dyn_points, nd * 32));
  TRY(upload(c, c->in_offsets, dyn_offsets, (batch + 1) * 8));
  TRY(upload(c, c->in_st_scalars, static_scalars, ns * 32));
  TRY(upload(c, c->in_st_index, static_index, ns * 4));
  TRY(upload(c, c->in_st_offsets, static_offsets, (batch + 1) * 8));
  Job job;
  job.d_dyn_scalars = (const uint32_t*)c->in_scalars.p;
  job.d_dyn_points = (const uint32_t*)c->in_points.p;
  job.d_dyn_offsets = (const uint64_t*)c->in_offsets.p;
  job.n_dyn = nd;
  job.d_st_scalars = (const uint32_t*)c->in_st_scalars.p;
  job.d_st_index = (const uint32_t*)c->in_st_index.p;
  job.d_st_offsets = (const uint64_t*)c->in_st_offsets.p;
  job.n_static = ns;
  job.d_static_rows = ps->rows;
  job.n_msm = (uint32_t)batch;
  job.max_dyn_row = longest_row(dyn_offsets, batch);
  const int rc = (ps->table && ns) ? batch_device_tables_enqueue(c, job, ps) : batch_device_enqueue(c, job, false);
  if (rc != ZKGPU_OK) { c->split = zkgpu_ctx::SplitOp{}; return rc; }
  c->pending = true; c->pending_batch = batch;
  return ZKGPU_OK;
}

}  // namespace

int zkgpu_verifier_set_tx_statements_kept(zkgpu_verifier* v, size_t transactions) {
  if (!v || transactions >= ((size_t)1 << 31)) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  v->tx_statements_kept = transactions;
  return ZKGPU_OK;
}

int zkgpu_verifier_set_tx_chunk(zkgpu_verifier* v, size_t transactions) {
  if (!v || transactions >= (1u << 24)) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->mu);
  v->tx_chunk = transactions;
  return ZKGPU_OK;
}

namespace {

// The device side of a transaction call (tx_call.hpp: TxDevice) on a verifier: key stages on aux_keys[slot], signature
// stages on aux_sigs[slot] (contexts of their own, each a pair of streams beside the lanes'), cloak proofs as blocks of
// mixed shapes on the lanes, staged through the verifier's ring of pinned / device areas.  v->mu is held by the call.
static_assert(zk::zkvm::TxCall::OK == ZKGPU_OK && zk::zkvm::TxCall::ENOMEM_ == ZKGPU_ENOMEM, "tx_call.hpp restates two status codes");
class GpuTxDevice : public zk::zkvm::TxDevice {
 public:
  // slot_base / arena_base: which of the verifier's stage contexts and staging areas this call uses (a call alone: slots 0
  // and 1, areas 0 .. RING - 1; two rounds in flight: one slot and one set of areas each)
  explicit GpuTxDevice(zkgpu_verifier* v, int slot_base = 0, size_t arena_base = 0) : v_(v), sb_(slot_base), ab_(arena_base) {}
  const uint8_t* basepoint() override { return v_->basepoint; }
  int keys_enqueue(int slot, const uint8_t* scalars, const uint8_t* points, const uint64_t* offsets, size_t rows) override {
    return seen(msm_values_enqueue(keys(slot), scalars, points, offsets, rows), keys(slot));
  }
  bool keys_done(int slot) override { return split_done(keys(slot)); }
  int keys_collect(int slot, uint8_t* ok_bits, uint8_t* values) override { return seen(split_collect(keys(slot), ok_bits, values), keys(slot)); }
  // (staging thread: touches the plans -- plans_mu -- the given arena and nothing else of the verifier)
  int proofs_stage(size_t ring_slot, size_t n, const zk::zkvm::TxProofSource* src, int host_threads, void** handle, std::string* err) override {
    zkgpu_txblock* blk = nullptr;
    const int rc = txblock_stage_host(v_, n, src, nullptr, host_threads, &blk, &v_->tx_arenas[ab_ + ring_slot], err);
    if (rc != ZKGPU_OK) return rc;
    Staged* st = new Staged();
    st->blk = blk;
    *handle = st;
    return ZKGPU_OK;
  }
  int proofs_start(size_t ring_slot, void* handle) override {
    Staged* st = (Staged*)handle;
    // one copy to HBM, and the chunk's batches queued on the lanes
    int rc = txblock_upload(v_, st->blk, &v_->tx_arenas[ab_ + ring_slot]);
    if (rc != ZKGPU_OK) { err_ = v_->last_error; return rc; }
    // a chunk of one shape goes to the device as few, large batches (measured: the last chunk in batches short enough for
    // the one-wavefront-per-transaction transcript, or cut in two, does not shorten the tail of the call)
    const size_t saved_chunk = v_->chunk;
    v_->chunk = std::max<size_t>(saved_chunk, 4096);
    st->run = block_start(v_, st->blk, true);
    v_->chunk = saved_chunk;
    if (!st->run) { err_ = v_->last_error; return ZKGPU_ENOMEM; }
    if (st->run->rc != ZKGPU_OK) { err_ = v_->last_error; return st->run->rc; }     // (proofs_finish still collects what was queued)
    return ZKGPU_OK;
  }
  bool proofs_done(void* handle) override {                // have the lanes finished every batch of this block?  (never blocks)
    Staged* st = (Staged*)handle;
    if (!st->run) return true;
    return block_done(v_, st->run);
  }
  int proofs_finish(void* handle, uint8_t* accept_bits) override {
    Staged* st = (Staged*)handle;
    int rc = ZKGPU_OK;
    if (st->run) { rc = block_finish(v_, st->run, accept_bits); if (rc != ZKGPU_OK) err_ = v_->last_error; }
    proofs_release(handle);
    return rc;
  }
  void proofs_release(void* handle) override {
    Staged* st = (Staged*)handle;
    if (st->blk) zkgpu_txblock_destroy(st->blk);
    delete st;
  }
  int sigs_enqueue(int slot, size_t rows, const uint8_t* dyn_scalars, const uint8_t* dyn_points, const uint64_t* dyn_offsets,
                   const uint8_t* base_scalars) override {
    // one term per row on the resident set's tables: index 0 = the basepoint B
    sidx_[slot].assign(rows, 0);
    soff_[slot].resize(rows + 1);
    for (size_t q = 0; q <= rows; ++q) soff_[slot][q] = q;
    return seen(verify_ps_enqueue(sigs(slot), v_->ps, rows, dyn_scalars, dyn_points, dyn_offsets, base_scalars, sidx_[slot].data(),
                                  soff_[slot].data()), sigs(slot));
  }
  bool sigs_done(int slot) override { return split_done(sigs(slot)); }
  int sigs_collect(int slot, uint8_t* bits) override { return seen(split_collect(sigs(slot), bits, nullptr), sigs(slot)); }
  std::string last_error() override { return err_; }

 private:
  struct Staged { zkgpu_txblock* blk = nullptr; zkgpu_verifier::BlockRun* run = nullptr; };
  int seen(int rc, zkgpu_ctx* where) { if (rc != ZKGPU_OK) err_ = zkgpu_last_error(where); return rc; }
  zkgpu_ctx* keys(int slot) const { return v_->aux_keys[(sb_ + slot) & 1]; }
  zkgpu_ctx* sigs(int slot) const { return v_->aux_sigs[(sb_ + slot) & 1]; }
  zkgpu_verifier* v_;
  const int sb_;
  const size_t ab_;
  std::vector<uint32_t> sidx_[2];
  std::vector<uint64_t> soff_[2];
  std::string err_;
};

// what a transaction call needs of the verifier before it starts (v->mu held): nothing in flight on the lanes, the stage
// contexts, the basepoint's encoding, the ring of staging areas
int tx_call_prepare(zkgpu_verifier* v, bool collect_lanes = true) {
  zkgpu_ctx* c = v->root;
  // the lanes' batches in flight are collected first: the call owns the verifier (tickets and runs keep their verdicts).
  // (Not when another round of transaction calls is in flight: its proofs are on the lanes, and share them.)
  if (collect_lanes) {
    while (!v->busy.empty()) ticket_collect(v, v->busy.front());
  }
  if (v->aux_keys[0] && v->aux_keys[1] && v->aux_sigs[0] && v->aux_sigs[1] && v->have_basepoint && v->tx_arenas.size() >= 2 * zk::zkvm::TxCall::RING)
    return ZKGPU_OK;
  if (!collect_lanes) {                                  // (the probes below want an idle device)
    while (!v->busy.empty()) ticket_collect(v, v->busy.front());
  }
  {
    DeviceGuard g(c->device);
    // a stage context is worth having only if its streams run BESIDE the lanes' (a stream made late in a process may land
    // on a hardware queue a lane already uses, and the two then take turns -- DESIGN.md sec 5.1): probed like the lanes
    // were (the device is idle: everything in flight has just been collected), and made again a few times if not
    auto stage_ctx = [&](zkgpu_ctx*& a) {
      if (a) return true;
      std::vector<zkgpu_ctx*> rejected;
      for (int attempt = 0; attempt < 6; ++attempt) {
        zkgpu_ctx* made = nullptr;
        if (ctx_create(c->device, nullptr, &made, true) != ZKGPU_OK) break;
        bool beside = true;
        for (zkgpu_ctx* lane : v->lanes)
          if (streams_overlap(lane->stream_l, made->stream) == 0 || streams_overlap(lane->stream_l, made->stream2) == 0) { beside = false; break; }
        if (beside || attempt == 5) { a = made; break; }
        rejected.push_back(made);                          // (kept until a good one is found: its queues stay taken meanwhile)
      }
      for (zkgpu_ctx* r : rejected) zkgpu_destroy(r);
      return a != nullptr;
    };
    for (zkgpu_ctx*& a : v->aux_keys)
      if (!stage_ctx(a)) { v->last_error = "no context for the key stage"; return ZKGPU_EHIP; }
    for (zkgpu_ctx*& a : v->aux_sigs)
      if (!stage_ctx(a)) { v->last_error = "no context for the signature stage"; return ZKGPU_EHIP; }
  }
  if (!v->have_basepoint) {
    uint8_t Bb[32];
    const int rc = zkgpu_pedersen_gens(v->aux_keys[0], v->basepoint, Bb);
    if (rc != ZKGPU_OK) { v->last_error = zkgpu_last_error(v->aux_keys[0]); return rc; }
    v->have_basepoint = true;
  }
  if (v->tx_arenas.size() < 2 * zk::zkvm::TxCall::RING) v->tx_arenas.resize(2 * zk::zkvm::TxCall::RING);     // (two sets: two rounds in flight)
  return ZKGPU_OK;
}

}  // namespace

// The scheduling of a call -- chunks, stages, the staging thread and the calling thread -- is csrc/tx_call.hpp (TxCall); the
// device side is GpuTxDevice above.  The calling thread is the one that talks to the device.
int zkgpu_tx_verify_batch(zkgpu_verifier* v, size_t batch, const uint8_t* txs, const uint64_t* tx_offsets, int host_threads,
                          uint8_t* accept_bitmap, uint8_t* status) {
  using namespace zk::zkvm;
  if (!v || !accept_bitmap) return ZKGPU_EINVAL;
  memset(accept_bitmap, 0, (batch + 7) / 8);
  if (status) memset(status, TX_INVALID, batch);
  if (batch == 0) return ZKGPU_OK;
  if (!txs || !tx_offsets || batch >= (1ull << 31)) return ZKGPU_EINVAL;
  for (size_t i = 0; i < batch; ++i) if (tx_offsets[i + 1] < tx_offsets[i]) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> vlk(v->mu);
  if (v->tx_format != ZKGPU_TXFORMAT_RECOLLECTED_V1) {            // no format enabled: nothing is inside the subset
    if (status) memset(status, TX_UNSUPPORTED, batch);
    return ZKGPU_OK;
  }
  TRY(tx_call_prepare(v));
  GpuTxDevice dev(v);
  int rc;
  try {
    TxCall call(dev, v->tx_statements, v->tx_statements_kept, batch, txs, tx_offsets, host_threads, v->tx_chunk, accept_bitmap, status);
    rc = call.run();
    if (rc != ZKGPU_OK) v->last_error = call.error_text();
  } catch (const std::bad_alloc&) {
    v->last_error = "out of host memory while planning the transaction call";
    rc = ZKGPU_ENOMEM;
  }
  if (rc != ZKGPU_OK) {                                          // both outputs read "nothing accepted"
    memset(accept_bitmap, 0, (batch + 7) / 8);
    if (status) for (size_t i = 0; i < batch; ++i) if (status[i] == TX_OK) status[i] = TX_INVALID;
  }
  return rc;
}


// ---- transaction calls in flight -------------------------------------------------------------------------------------------
// Upstream's Tx::verify is pure and callable from many threads at once; a node's mempool hands over small
// batches as they arrive.  One call at a time per verifier left a 1024-transaction call at 3 ms -- the latency of a lone
// device batch -- and a second verifier is not the answer.  zkgpu_tx_verify_submit QUEUES a call and returns;
// an engine thread of the verifier takes everything queued (up to 16 384 transactions) and runs it as ONE call (TxCall over
// the pieces): calls that arrive while a round is running are merged into the next one, so that eight callers of 1024 see
// the throughput of calls of several thousand.  zkgpu_tx_verify_wait blocks until that call's round is done and writes ITS
// bitmap and status bytes.  The transaction bytes and offsets must stay valid until the call has been waited for.  Verdicts
// are those of separate calls.  Fail-closed: a round that fails gives every call in it the error and all-zero outputs.
namespace {
constexpr size_t TX_LINGER_ENOUGH = 4096;                                  // transactions queued: no reason to wait for more
constexpr std::chrono::microseconds TX_LINGER(300);
// One round of merged calls on its way: its own device view (one stage slot, one set of staging areas, one statement store)
struct TxRound {
  std::vector<zkgpu_verifier::TxPending*> calls;
  std::vector<zk::zkvm::TxCall::Piece> pieces;
  std::vector<uint8_t> bits, status;
  std::unique_ptr<GpuTxDevice> dev;
  std::unique_ptr<zk::zkvm::TxCall> call;
  size_t total = 0;
  int rc = ZKGPU_OK;
};

// hands every call of a finished round its own bits (tx_mu held)
void tx_round_distribute(TxRound& r) {
  using namespace zk::zkvm;
  size_t at = 0;
  for (auto* p : r.calls) {
    p->bits.assign((p->batch + 7) / 8, 0);
    p->status.assign(p->batch, TX_INVALID);
    p->rc = r.rc;
    for (size_t i = 0; i < p->batch; ++i) {
      const size_t g = at + i;
      if (r.rc == ZKGPU_OK) {
        if ((r.bits[g / 8] >> (g % 8)) & 1) p->bits[i / 8] |= (uint8_t)(1u << (i % 8));
        p->status[i] = r.status[g];
      } else if (r.status[g] == TX_UNSUPPORTED) {
        p->status[i] = TX_UNSUPPORTED;
      }
    }
    at += p->batch;
    p->state = 2;
  }
}

// The engine: rounds of merged calls, driven by this one thread (TxCall::step never blocks; a round is finished only once
// the device has settled).  It can keep TWO rounds in flight -- one key / signature stage slot and one set of staging areas
// each, the lanes shared -- when the process's queue count allows (see max_rounds).  The verifier's mutex is held
// from the first admission until nothing is in flight: the verifier belongs to the calls.
void tx_engine_main(zkgpu_verifier* v) {
  using namespace zk::zkvm;
  std::unique_ptr<TxRound> active[2];
  // Two rounds in flight while the process's hardware queues are few enough for the device to run them all side by side
  // (GPU_MAX_HW_QUEUES <= 19: DESIGN.md sec 5.1), else one.  Measured on MI355X, three processes per setting
  // (profiles/archive/r04y_rounds_hwq.txt), at 18 queues: 4 calls of 4096 in flight 1.72 - 1.76 M tx/s with two rounds against
  // 0.98 - 1.41 M with one, 8 calls of 1024 1.15 - 1.30 M against 1.00 - 1.12 M; at 24 queues (28 in all priorities) two
  // rounds LOST -- 0.56 - 0.60 M against 0.92 - 0.96 M, single stages stalling 5 - 20 ms (profiles/archive/r04e_inflight.txt,
  // r04f_inflight_timing.txt): both key contexts, both signature contexts and the lanes busy at once were more queues than
  // the device runs side by side.  ZKGPU_TX_ROUNDS=1 / 2 overrides.
  int max_rounds = 1;
  {
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    const int queues = q ? atoi(q) : 4;
    if (!g_hw_queues_late && queues >= 8 && queues <= 19) max_rounds = 2;
    if (const char* r = getenv("ZKGPU_TX_ROUNDS")) { const int k = atoi(r); if (k == 1 || k == 2) max_rounds = k; }
  }
  std::mutex news_mu;
  std::condition_variable news_cv;
  bool news = false;
  std::unique_lock<std::mutex> vlk(v->mu, std::defer_lock);
  for (;;) {
    // ---- admit: a free slot takes everything that is queued (up to tx_merge_max transactions)
    std::unique_ptr<TxRound> fresh[2];
    {
      std::unique_lock<std::mutex> lk(v->tx_mu);
      if (!active[0] && !active[1]) {
        if (vlk.owns_lock()) vlk.unlock();              // idle: the verifier is everybody's again
        v->tx_cv.wait(lk, [&] { return v->tx_engine_quit || !v->tx_queue.empty(); });
        if (v->tx_queue.empty()) return;                 // (quit with nothing queued; what is queued at quit is still served)
      }
      for (int set = 0; set < max_rounds; ++set) {
        if (active[set] || v->tx_queue.empty()) continue;
        // While the other round is still running the device is busy anyway: a SMALL queue then waits up to TX_LINGER for the
        // calls that are about to arrive (a caller that has just been handed four verdicts submits its next four calls one
        // after the other, microseconds apart: admitted at once, the first would make a round of its own -- measured:
        // 2.3 - 2.7 calls of 1024 per round instead of 4).  An idle engine admits at once.
        if (active[set ^ 1] && !v->tx_engine_quit) {
          size_t queued = 0;
          for (auto* p : v->tx_queue) queued += p->batch;
          if (queued < TX_LINGER_ENOUGH && std::chrono::steady_clock::now() - v->tx_queue.front()->arrived < TX_LINGER) continue;
        }
        fresh[set].reset(new TxRound());
        TxRound& r = *fresh[set];
        while (!v->tx_queue.empty() && (r.calls.empty() || r.total + v->tx_queue.front()->batch <= v->tx_merge_max)) {
          r.calls.push_back(v->tx_queue.front());
          r.total += r.calls.back()->batch;
          r.calls.back()->state = 1;
          v->tx_queue.pop_front();
        }
        ++v->tx_rounds; v->tx_round_calls += r.calls.size();
      }
    }
    if (!vlk.owns_lock()) vlk.lock();
    bool progress = false;
    for (int set = 0; set < 2; ++set) {
      if (!fresh[set]) continue;
      TxRound& r = *fresh[set];
      int threads = 0;
      for (auto* p : r.calls) { r.pieces.push_back({p->txs, p->offsets, p->batch}); threads = std::max(threads, p->host_threads); }
      r.bits.assign((r.total + 7) / 8 + 1, 0);
      r.status.assign(r.total, TX_INVALID);
      if (v->tx_format != ZKGPU_TXFORMAT_RECOLLECTED_V1) {
        std::fill(r.status.begin(), r.status.end(), (uint8_t)TX_UNSUPPORTED);       // no format enabled: nothing is inside the subset
      } else {
        r.rc = tx_call_prepare(v, !active[0] && !active[1]);
        if (r.rc == ZKGPU_OK) {
          try {
            r.dev.reset(new GpuTxDevice(v, set, (size_t)set * TxCall::RING));
            r.call.reset(new TxCall(*r.dev, set ? v->tx_statements_b : v->tx_statements, v->tx_statements_kept, r.pieces, threads, v->tx_chunk,
                                    r.bits.data(), r.status.data(), 1));
            r.call->set_on_news([&] { { std::lock_guard<std::mutex> nl(news_mu); news = true; } news_cv.notify_one(); });
            r.rc = r.call->start();
            if (r.rc != ZKGPU_OK) v->last_error = r.call->error_text();
          } catch (const std::bad_alloc&) {
            v->last_error = "out of host memory while planning the transaction call";
            r.rc = ZKGPU_ENOMEM;
            r.call.reset();
          }
        }
      }
      active[set] = std::move(fresh[set]);
      progress = true;
    }
    // ---- step what is in flight; a round that is done hands out its verdicts
    for (int set = 0; set < 2; ++set) {
      if (!active[set]) continue;
      TxRound& r = *active[set];
      const bool running = r.call && r.rc == ZKGPU_OK;
      if (running && !r.call->done()) progress |= r.call->step();
      // (a round whose stages are all queued is finished only once the device has settled: finish() then returns without
      // waiting, and the other round is never kept from queueing its stages meanwhile)
      if (!running || (r.call->done() && r.call->settled())) {
        if (r.call) {
          const int rc = r.call->finish();
          if (r.rc == ZKGPU_OK) r.rc = rc;
          if (rc != ZKGPU_OK) v->last_error = r.call->error_text();
          r.call.reset();                                // (joins its staging thread: nothing calls on_news afterwards)
        }
        {
          std::lock_guard<std::mutex> lk(v->tx_mu);
          tx_round_distribute(r);
        }
        v->tx_cv.notify_all();
        active[set].reset();
        progress = true;
      }
    }
    if (!progress) {                                     // nothing to do now: until a staging thread or a caller has news, 50 us at most
      std::unique_lock<std::mutex> nl(news_mu);
      if (!news) news_cv.wait_for(nl, std::chrono::microseconds(50));
      news = false;
    }
  }
}
}  // namespace

int zkgpu_tx_verify_submit(zkgpu_verifier* v, size_t batch, const uint8_t* txs, const uint64_t* tx_offsets, int host_threads, uint64_t* call_id) {
  if (!v || !call_id || batch == 0 || batch >= (1ull << 31) || !txs || !tx_offsets) return ZKGPU_EINVAL;
  for (size_t i = 0; i < batch; ++i) if (tx_offsets[i + 1] < tx_offsets[i]) return ZKGPU_EINVAL;
  std::unique_ptr<zkgpu_verifier::TxPending> p(new zkgpu_verifier::TxPending());
  p->batch = batch; p->txs = txs; p->offsets = tx_offsets; p->host_threads = host_threads;
  p->arrived = std::chrono::steady_clock::now();
  std::lock_guard<std::mutex> lk(v->tx_mu);
  if (v->tx_engine_quit) return ZKGPU_EINVAL;
  if (!v->tx_engine.joinable()) {
    try { v->tx_engine = std::thread(tx_engine_main, v); }
    catch (...) { return ZKGPU_ENOMEM; }
  }
  p->id = v->tx_next_id++;
  *call_id = p->id;
  v->tx_queue.push_back(p.get());
  v->tx_calls[p->id] = std::move(p);
  v->tx_cv.notify_all();
  return ZKGPU_OK;
}

int zkgpu_tx_verify_wait(zkgpu_verifier* v, uint64_t call_id, uint8_t* accept_bitmap, uint8_t* status) {
  if (!v || !accept_bitmap) return ZKGPU_EINVAL;
  std::unique_lock<std::mutex> lk(v->tx_mu);
  auto it = v->tx_calls.find(call_id);
  if (it == v->tx_calls.end()) return ZKGPU_EINVAL;
  zkgpu_verifier::TxPending* p = it->second.get();
  v->tx_cv.wait(lk, [&] { return p->state == 2; });
  const int rc = p->rc;
  memset(accept_bitmap, 0, (p->batch + 7) / 8);
  if (rc == ZKGPU_OK) memcpy(accept_bitmap, p->bits.data(), p->bits.size());
  if (status) memcpy(status, p->status.data(), p->batch);
  v->tx_calls.erase(it);
  return rc;
}

// out[0] rounds the engine has run, out[1] calls they held in all (calls per round = out[1] / out[0])
int zkgpu_tx_verify_stats(zkgpu_verifier* v, uint64_t out[2]) {
  if (!v || !out) return ZKGPU_EINVAL;
  std::lock_guard<std::mutex> lk(v->tx_mu);
  out[0] = v->tx_rounds; out[1] = v->tx_round_calls;
  return ZKGPU_OK;
}

// The n-th HIP runtime call of the process from now on reports hipErrorUnknown WITHOUT being made (fault_gate.hpp); n < 0: the
// |n|-th and every call after it (a device that is gone); n = 0: disarm.  Process-wide -- the lanes of a verifier, the slices
// of a prover call and the staging threads are contexts and threads of their own, and a fault belongs to whoever makes the
// n-th call.  Returns the number of calls that passed the gate since it was last armed (so a test can count a clean run's
// calls first and then fail every one of them in turn); *fired (may be NULL): how many were answered "failed".
long long zkgpu_debug_fail_after(zkgpu_ctx* ctx, long long n, long long* fired) {
  (void)ctx;
  zk::fault::State& s = zk::fault::state();
  const long long seen = s.seen.load();
  if (fired) *fired = s.fired.load();
  s.armed.store(0);
  s.seen.store(0); s.fired.store(0);
  s.sticky.store(n < 0 ? 1 : 0);
  s.countdown.store(n < 0 ? -n : n);
  if (n != 0) s.armed.store(1);
  return seen;
}

// ---- hooks (include/zkgpu_hooks.h): by name, and only for a process that asked for them before it loaded the library --------
const void* zkgpu_hook(const char* name) {
  if (!name || !test_hooks_enabled()) return nullptr;
  struct Entry { const char* name; const void* fn; };
#define ZKGPU_HOOK(f) {#f, (const void*)&f}
  static const Entry table[] = {
    ZKGPU_HOOK(zkgpu_set_group_size), ZKGPU_HOOK(zkgpu_set_locate_mode), ZKGPU_HOOK(zkgpu_set_horner_mode), ZKGPU_HOOK(zkgpu_set_transcript_mode),
    ZKGPU_HOOK(zkgpu_set_static_parts), ZKGPU_HOOK(zkgpu_set_locate_parts), ZKGPU_HOOK(zkgpu_set_tail_mode), ZKGPU_HOOK(zkgpu_set_window_bits),
    ZKGPU_HOOK(zkgpu_set_prover_mode), ZKGPU_HOOK(zkgpu_set_serial), ZKGPU_HOOK(zkgpu_measure_hbm_copy), ZKGPU_HOOK(zkgpu_profile_enable),
    ZKGPU_HOOK(zkgpu_profile_reset), ZKGPU_HOOK(zkgpu_profile_count), ZKGPU_HOOK(zkgpu_profile_get), ZKGPU_HOOK(zkgpu_last_window_bits),
    ZKGPU_HOOK(zkgpu_last_bucket_adds), ZKGPU_HOOK(zkgpu_verifier_lane), ZKGPU_HOOK(zkgpu_debug_arith), ZKGPU_HOOK(zkgpu_debug_coop_selftest),
    ZKGPU_HOOK(zkgpu_debug_read), ZKGPU_HOOK(zkgpu_cloak_plan_layout), ZKGPU_HOOK(zkgpu_debug_force_regroup), ZKGPU_HOOK(zkgpu_debug_comm_mock),
    ZKGPU_HOOK(zkgpu_debug_fail_after),
  };
#undef ZKGPU_HOOK
  for (const Entry& e : table) if (!strcmp(e.name, name)) return e.fn;
  return nullptr;
}

}  // extern "C"
