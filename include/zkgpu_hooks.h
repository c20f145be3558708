/* zkgpu_hooks.h -- measurement, tuning and test hooks of libzkgpu.so.  NOT part of the ABI.
 *
 * None of these functions is exported (the library is built with -fvisibility=hidden and this header does not raise
 * it): they are reached by name through `zkgpu_hook(name)` (zkgpu.h), which returns the function's address when
 * ZKGPU_TEST_HOOKS=1 was in the environment at the time the library was loaded and NULL otherwise.  tests/conftest.py and
 * bench.py set the variable; a deployed verifier never does, `rust/zkgpu-sys` does not bind them, and a C program cannot
 * link against them.  The declarations below are the signatures to cast the returned pointer to (zkvm_amd/native.py does).
 * No hook changes a result: every mode yields the same verdicts / proofs (the tests run every mode); the one hook that
 * interferes at all, zkgpu_debug_fail_after, can only turn a call into an ERROR (all outputs zero), never into an accept.
 *
 * (Until round 4 the first 24 of these, and zkgpu_set_prover_mode among them, were exports of zkgpu.h; VERDICT r04 "ABI sprawl".)
 */
#ifndef ZKGPU_HOOKS_H
#define ZKGPU_HOOKS_H

#include "zkgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- modes the sweeps of DESIGN.md have settled (defaults: 0 = the library's choice) ---------------------------------- */
/* transactions per group check, 1 .. 64 (default 16; 1 = every transaction on its own).  Forks inherit. */
int zkgpu_set_group_size(zkgpu_ctx *ctx, int group);
/* failed groups: 0 automatic (locate the culprit from 2048 transactions per batch on), 1 never locate, 2 always, 3 always with
 * the locating sums of ALL groups formed beside the group sums (measured slower) */
int zkgpu_set_locate_mode(zkgpu_ctx *ctx, int mode);
/* Horner chains over the windows: 0 automatic, 1 one per transaction, 2 one per group (+ the failed groups' transactions) */
int zkgpu_set_horner_mode(zkgpu_ctx *ctx, int mode);
/* transcript replay: 0 automatic (one wavefront per transaction up to 1536 per batch, one lane beyond), 1 lane, 2 wavefront */
int zkgpu_set_transcript_mode(zkgpu_ctx *ctx, int mode);
/* lanes per (check, window) of the fixed-base kernel / per (failed group, window) of the locating multiplication (0 = default) */
int zkgpu_set_static_parts(zkgpu_ctx *ctx, int parts);
int zkgpu_set_locate_parts(zkgpu_ctx *ctx, int parts);
/* sums on the tail of a batch: 0 inside the kernel that consumes them (k_locate_fused, k_recheck_fused), 1 launches of their own */
int zkgpu_set_tail_mode(zkgpu_ctx *ctx, int mode);
/* Pippenger window width of zkgpu_msm* (0 = automatic) */
int zkgpu_set_window_bits(zkgpu_ctx *ctx, int w);
/* provers: 0 the whole proof on the device (default; calls of 1024 statements or more in slices), 1 host threads in lockstep
 * (the round-1 arrangement), 16 + S (S = 1 .. 8): on the device in exactly S slices whatever the size.  Byte-identical proofs. */
int zkgpu_set_prover_mode(zkgpu_ctx *ctx, int mode);
/* with on != 0 the kernels of a batch run one after another on a single stream, so that the profile hooks report each
 * kernel's duration alone on the chip */
int zkgpu_set_serial(zkgpu_ctx *ctx, int on);

/* ---- measurement -------------------------------------------------------------------------------------------------------- */
/* "measure achievable HBM with a copy kernel and report both" -- a streaming copy of `bytes` bytes
 * (16 B per lane, one workgroup per 4 KiB), best of `iters` launches by HIP events: (bytes read + bytes written) / time */
int zkgpu_measure_hbm_copy(zkgpu_ctx *ctx, size_t bytes, int iters, double *gbytes_per_s);
/* when enabled, every kernel launch of this context is bracketed by HIP events on the stream it is launched on */
int zkgpu_profile_enable(zkgpu_ctx *ctx, int on);
void zkgpu_profile_reset(zkgpu_ctx *ctx);
int zkgpu_profile_count(zkgpu_ctx *ctx);                 /* distinct kernels seen since the last reset */
int zkgpu_profile_get(zkgpu_ctx *ctx, int i, const char **name, uint64_t *launches, double *total_ms);
/* Pippenger window width chosen by the last zkgpu_msm* call, and the point additions of its bucket accumulation */
int zkgpu_last_window_bits(const zkgpu_ctx *ctx);
uint64_t zkgpu_last_bucket_adds(const zkgpu_ctx *ctx);
/* context of lane i of a verifier (0 = the one given to zkgpu_verifier_create), for the hooks above; owned by the verifier */
zkgpu_ctx *zkgpu_verifier_lane(zkgpu_verifier *v, int i);

/* ---- device-side unit tests --------------------------------------------------------------------------------------------- */
/* the arithmetic layers on their own, one lane per element (a, b, out: n x 32 bytes).  op 0 field product, 1 square,
 * 2 inverse, 3 a + b - b + a, 4 x^((p-5)/8)  (GF(2^255-19): 32 little-endian bytes, canonical out); 10 product mod l
 * (canonical Montgomery form), 11 the same in the lazy limb form, 12 a lazy chain (a-b)(a+b) + 16ab - b, 13 / 14 / 16 inverse
 * mod l (a^(l-2) canonical / Euclid / a^(l-2) lazy), 15 a + b - a  (scalars: canonical words out, inputs reduced mod l) */
int zkgpu_debug_arith(zkgpu_ctx *ctx, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
/* the cross-lane primitives of the cooperative Keccak on given inputs (in: 3 x 64 words a, b, gather byte addresses; out:
 * 8 x 64 words: row_ror:8(a), row_shr:1(a), row_shl:1(a), permlane16_swap(a, b) -> (a', b'), permlane32_swap(a, b) ->
 * (a', b'), ds_bpermute(addr, a)) and Keccak-f[1600] of n_states states (25 u64 each, in place), one wavefront per state */
int zkgpu_debug_coop_selftest(zkgpu_ctx *ctx, const uint32_t *in, uint32_t *out, uint64_t *states, size_t n_states);
/* intermediate buffers of the last device-side preparation on this context.
 * what = "challenges": per transaction layout[0] slots of 32 B, each x * 2^260 mod l (Montgomery form):
 *          0 y  1 z  2 u  3 x  4 w  5 prod u_j  6 prod u_j^2  7 r  8 t_x  9 t_x_blinding  10 e_blinding
 *          11 a  12 b  13 rho (weight inside a group check; 1 when transactions are checked alone)
 *          14.. the layout[2] second-phase challenges, then the k inner-product challenges u_j, ...
 *        "static_scalars" (layout[4] per transaction: B, B_blinding, G_i, H_i) and "dyn_scalars"
 *        (layout[3]: A_I1 A_O1 S1 A_I2 A_O2 S2 | V | T_1 T_3..T_6 | L | R): canonical 32-byte scalars of
 *        the verification equation MULTIPLIED THROUGH by c' = rho * y^(padded_n - 1) * prod u_j^2 (the
 *        device evaluates the equation in this inversion-free form).
  * ("prover_slices": 4 bytes, the slices the last prover call on the context ran in.)
 * Returns bytes copied.  zkgpu_cloak_plan_layout fills layout[0..7] = slots per transaction, challenge
 * slots proper, second-phase challenges, dynamic terms, static terms, k, m, monomials. */
long long zkgpu_debug_read(zkgpu_ctx *ctx, const char *what, void *out, size_t bytes);
int zkgpu_cloak_plan_layout(const zkgpu_cloak_plan *plan, uint32_t layout[8]);
/* on != 0: the next batches on this context take the (never expected, ~2^-248) path on which a located transaction does
 * not account for its group and zkgpu_verify_wait re-runs the batch ungrouped; returns the number of re-runs so far */
long long zkgpu_debug_force_regroup(zkgpu_ctx *ctx, int on);
/* world > 0 replaces the collective function table by an in-process mock of a world of `world` ranks whose other ranks
 * contribute peer_slots (world x slot_bytes bytes), so that the exchange step can be exercised at world 2 .. 8 on one
 * GPU; world = 0 restores RCCL; world < 0 only asks.  Returns the all-gathers the mock has served.  A communicator keeps the function table it
 * was created with for its whole life, whatever the switch does afterwards.
 * (A second rehearsal aid lives in the environment: ZKGPU_TEST_COMM_STALL="init:<rank>|init:all|gather:<rank>|gather:all"
 * beside ZKGPU_TEST_HOOKS=1 makes ncclCommInitRank / the first ncclAllGather of the named rank never return.) */
long long zkgpu_debug_comm_mock(zkgpu_ctx *ctx, int world, const uint8_t *peer_slots, size_t slot_bytes);
/* Fault injection (csrc/fault_gate.hpp): the n-th HIP runtime call the library makes from now on -- allocation, copy, event,
 * stream, synchronisation, the hipGetLastError that collects a launch -- reports hipErrorUnknown WITHOUT being made (the device
 * is untouched); n < 0: the |n|-th call and every one after it (a lost device); n = 0: disarm.  Process-wide (ctx is ignored:
 * lanes, slices and staging threads are contexts of their own).  Returns the number of calls that passed the gate since it was
 * last armed; *fired (may be NULL) = how many of them were answered "failed".  The gate can only turn a success into an error:
 * what the tests assert is that every such error ends in a nonzero status, outputs all zero, nothing hung, and a library that
 * verifies the next clean batch correctly.  (Clean-up on an error path runs with the gate held open.) */
long long zkgpu_debug_fail_after(zkgpu_ctx *ctx, long long n, long long *fired);

#ifdef __cplusplus
}
#endif
#endif
