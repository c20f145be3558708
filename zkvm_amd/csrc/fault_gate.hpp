// fault_gate.hpp -- every HIP runtime call of the library's host code passes a gate that a test can make answer "failed".
//
// A consensus verifier's first property is that an error is never an accept and never a hang (SURVEY.md sec 5 "fail-closed",
// sec 8(b) zkgpu_verify_batch convention).  The error branches of the ticket engine, the sharded exchange and the sliced
// prover cannot be reached by asking the device nicely; they are reached by making the n-th runtime call of the process
// REPORT a failure.  The call itself is then NOT made -- the device is untouched, nothing leaks --
// and the caller sees hipErrorUnknown exactly where a real fault would have surfaced: a refused allocation, a copy or an
// event that could not be queued, a launch error collected by hipGetLastError, a failed synchronisation.
//
// Armed only through the hook zkgpu_debug_fail_after (include/zkgpu_hooks.h; NULL unless the process asked for the hooks
// before it loaded the library).  Unarmed the gate is one relaxed load per runtime call.  A gate can only turn a success
// into an error, and every error path ends in "bits zero": it cannot make the library accept anything.
//
// Included right after <hip/hip_runtime.h> and before any host code of the library: the function-like macros below shadow
// the runtime's names for everything that follows (a macro's own name is not expanded again inside its replacement).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>

namespace zk { namespace fault {

struct State {
  std::atomic<int> armed{0};
  std::atomic<long long> countdown{0};   // the check that brings it to zero fails
  std::atomic<long long> seen{0};        // checks passed through the gate since it was armed
  std::atomic<long long> fired{0};       // checks answered "failed"
  std::atomic<int> sticky{0};            // once fired, every later check fails too (a lost device)
};
inline State& state() { static State s; return s; }

// clean-up on an error path (synchronise what was queued, release an area) runs with the gate held open: the fault under
// test is the ONE that brought the code here
struct Suppress {
  static int& depth() { static thread_local int d = 0; return d; }
  Suppress() { ++depth(); }
  ~Suppress() { --depth(); }
};

inline hipError_t gate() {
  State& s = state();
  if (!s.armed.load(std::memory_order_relaxed)) return hipSuccess;
  if (Suppress::depth()) return hipSuccess;
  s.seen.fetch_add(1, std::memory_order_relaxed);
  if (s.countdown.fetch_sub(1, std::memory_order_relaxed) == 1) { s.fired.fetch_add(1); return hipErrorUnknown; }
  if (s.sticky.load(std::memory_order_relaxed) && s.fired.load(std::memory_order_relaxed)) { s.fired.fetch_add(1); return hipErrorUnknown; }
  return hipSuccess;
}

}}  // namespace zk::fault

// (GNU `a ?: b`: a when it is non-zero, else b -- the real call is evaluated only when the gate answers hipSuccess)
// hipEventQuery / hipStreamQuery are NOT gated: they are polls in spin loops (how many are made depends on timing, which would make
// the n-th call of a run mean something else every time), they change nothing, and whatever they answer the collecting call that
// follows synchronises for real -- that one is gated.
#define hipMalloc(...) ((hipError_t)(zk::fault::gate() ?: hipMalloc(__VA_ARGS__)))
#define hipHostMalloc(...) ((hipError_t)(zk::fault::gate() ?: hipHostMalloc(__VA_ARGS__)))
#define hipMemcpy(...) ((hipError_t)(zk::fault::gate() ?: hipMemcpy(__VA_ARGS__)))
#define hipMemcpyAsync(...) ((hipError_t)(zk::fault::gate() ?: hipMemcpyAsync(__VA_ARGS__)))
#define hipMemsetAsync(...) ((hipError_t)(zk::fault::gate() ?: hipMemsetAsync(__VA_ARGS__)))
#define hipEventCreate(...) ((hipError_t)(zk::fault::gate() ?: hipEventCreate(__VA_ARGS__)))
#define hipEventCreateWithFlags(...) ((hipError_t)(zk::fault::gate() ?: hipEventCreateWithFlags(__VA_ARGS__)))
#define hipEventRecord(...) ((hipError_t)(zk::fault::gate() ?: hipEventRecord(__VA_ARGS__)))
#define hipEventSynchronize(...) ((hipError_t)(zk::fault::gate() ?: hipEventSynchronize(__VA_ARGS__)))
#define hipStreamCreate(...) ((hipError_t)(zk::fault::gate() ?: hipStreamCreate(__VA_ARGS__)))
#define hipStreamCreateWithFlags(...) ((hipError_t)(zk::fault::gate() ?: hipStreamCreateWithFlags(__VA_ARGS__)))
#define hipStreamCreateWithPriority(...) ((hipError_t)(zk::fault::gate() ?: hipStreamCreateWithPriority(__VA_ARGS__)))
#define hipStreamSynchronize(...) ((hipError_t)(zk::fault::gate() ?: hipStreamSynchronize(__VA_ARGS__)))
#define hipStreamWaitEvent(...) ((hipError_t)(zk::fault::gate() ?: hipStreamWaitEvent(__VA_ARGS__)))
#define hipDeviceSynchronize(...) ((hipError_t)(zk::fault::gate() ?: hipDeviceSynchronize(__VA_ARGS__)))
#define hipGetLastError(...) ((hipError_t)(zk::fault::gate() ?: hipGetLastError(__VA_ARGS__)))
