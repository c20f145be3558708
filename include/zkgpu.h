/* zkgpu.h -- C ABI of libzkgpu.so, the MI355X (gfx950) back end for the
 * multiscalar-multiplication tail of ZkVM / Bulletproofs-R1CS verification.
 *
 * What each entry point replaces in the reference (interstellar/zkvm ->
 * slingshot/zkvm; /root/reference holds only README.md:1-7, so the items below
 * are named by their upstream Rust paths):
 *
 *   zkgpu_msm*            RistrettoPoint::vartime_multiscalar_mul /
 *                         optional_multiscalar_mul  (curve25519-dalek,
 *                         traits::VartimeMultiscalarMul) followed by
 *                         RistrettoPoint::compress
 *   zkgpu_verify_batch*   the tail of bulletproofs::r1cs::Verifier::verify
 *                         (`mega_check = optional_multiscalar_mul(..)`,
 *                         `mega_check.is_identity()`) for a batch of
 *                         independent proofs, one accept bit per proof
 *   zkgpu_pointset_*      BulletproofGens / PedersenGens held decompressed on
 *                         the device (the reference keeps them decompressed
 *                         in host memory)
 *
 * Conventions
 *   - plain pointers and sizes, no allocation handed across, no callbacks;
 *   - scalars: 32 bytes little endian, bit 255 clear (dalek `Scalar` invariant);
 *     points: 32-byte ristretto255 encodings (RFC 9496 sec 4.3.1);
 *   - every call returns ZKGPU_OK (0) or a negative error code; on ANY error
 *     the outputs are zeroed (fail-closed: an error is never an "accept");
 *   - a context is bound to one GPU, owns its HIP streams (three chip-filling ones, shared with
 *     its forks, and a light one of its own) and serialises its own calls; use one context per
 *     thread, one process per GPU (zkgpu_comm joins the processes of a node over RCCL);
 *   - `*_dev` variants take device pointers (inputs already resident in HBM) and
 *     are what bench.py times; host-pointer variants add the PCIe copies;
 *   - a context with a SUBMITTED batch (zkgpu_*_submit*, not yet collected by zkgpu_verify_wait) owns its
 *     workspace and result buffers on behalf of that batch: until then every synchronous entry point called
 *     on it returns ZKGPU_EINVAL with its outputs zeroed (it never runs over the batch in flight).
 */
#ifndef ZKGPU_H
#define ZKGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: what this header declares is what it exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define ZKGPU_OK 0
#define ZKGPU_EINVAL (-1)          /* bad argument (null pointer, bad offsets, scalar bit 255 set) */
#define ZKGPU_EINVALID_POINT (-2)  /* zkgpu_msm*: a point failed RFC 9496 DECODE */
#define ZKGPU_EHIP (-3)            /* HIP runtime error; see zkgpu_last_error */
#define ZKGPU_ENOMEM (-4)
#define ZKGPU_ENODEVICE (-5)
#define ZKGPU_ENOCOMM (-6)         /* RCCL could not be loaded / initialised, or a collective failed */
#define ZKGPU_EREMOTE (-7)         /* sharded verification: another rank reported an error (bitmap zeroed everywhere) */
/* the one POSITIVE status: the call succeeded and its result is usable, but the caller should know something */
#define ZKGPU_WSECOND_VERIFIER 1   /* zkgpu_verifier_create: another verifier is alive on this device in this process while the
                                      runtime hands out 20 or more hardware queues: their streams together oversubscribe the
                                      device's queue slots and calls take 5 - 40 ms instead of a steady 6.
                                      The verifier IS created (*out is valid).  Use ONE verifier per process and device --
                                      tickets, blocks and transaction calls of any mix go through one -- or export
                                      GPU_MAX_HW_QUEUES=16 before the process's first HIP call. */

typedef struct zkgpu_ctx zkgpu_ctx;
typedef struct zkgpu_pointset zkgpu_pointset;

/* ABI version of this header (bumped on any signature change). */
int zkgpu_abi_version(void);
const char *zkgpu_strerror(int code);
/* Human-readable detail of the last failing HIP call on this context. */
const char *zkgpu_last_error(const zkgpu_ctx *ctx);

/* The library never edits the process environment.  What it needs from it is ONE HIP runtime variable that the runtime
 * reads when it starts: call this BEFORE the process's first HIP call (before zkgpu_init, before anything else that touches
 * the GPU) and export what it writes into buf -- "NAME=value", NUL-terminated, today "GPU_MAX_HW_QUEUES=18" (why 18:
 * DESIGN.md sec 5.1) -- with the host language's own setenv.  Returns
 *   ZKGPU_HINT_APPLY   (0) the variable is unset and the runtime has not started: export it now, it will count;
 *   ZKGPU_HINT_PRESENT (1) the caller has set the variable already: nothing to do (buf still holds the recommendation);
 *   ZKGPU_HINT_LATE    (2) the runtime is up without it: exporting changes nothing now; the process runs on the runtime's
 *                          default of 4 queues, verdicts are unaffected, batches in flight take turns; remembered and
 *                          reported by zkgpu_ctx_queue_info / zkgpu_verifier_queue_info / zkgpu_verifier_last_error.
 * A process that never calls it and never sets the variable is treated as LATE by its first zkgpu_init.
 * (Replaces the setenv zkgpu_init used to do itself: an embedding application's environment is not the library's to edit.) */
#define ZKGPU_HINT_APPLY 0
#define ZKGPU_HINT_PRESENT 1
#define ZKGPU_HINT_LATE 2
int zkgpu_runtime_hint(char *buf, size_t cap);

/* Create a context on HIP device `device`. */
int zkgpu_init(int device, zkgpu_ctx **out);
void zkgpu_destroy(zkgpu_ctx *ctx);

/* out = compress(sum_i scalars[i] * decompress(points[i])), n >= 0.
 * An undecodable point gives ZKGPU_EINVALID_POINT and *bad_index = the lowest
 * offending index (bad_index may be NULL). */
int zkgpu_msm(zkgpu_ctx *ctx, const uint8_t *scalars, const uint8_t *points, size_t n,
              uint8_t out[32], size_t *bad_index);
int zkgpu_msm_dev(zkgpu_ctx *ctx, const void *d_scalars, const void *d_points, size_t n,
                  uint8_t out[32], size_t *bad_index);

/* Batch of `batch` independent checks  sum_j s_ij * P_ij == identity,  rows of a
 * CSR layout: check i owns terms offsets[i] .. offsets[i+1]-1.  Bit i of
 * accept_bitmap (byte i/8, bit i%8) is 1 iff every point of check i decodes and
 * the sum is the identity.  accept_bitmap has ceil(batch/8) bytes (host memory
 * in both variants). */
int zkgpu_verify_batch(zkgpu_ctx *ctx, const uint8_t *scalars, const uint8_t *points,
                       const uint64_t *offsets, size_t batch, uint8_t *accept_bitmap);
int zkgpu_verify_batch_dev(zkgpu_ctx *ctx, const void *d_scalars, const void *d_points,
                           const void *d_offsets, size_t batch, size_t n_terms,
                           uint8_t *accept_bitmap);

/* A set of points decompressed once and kept on the device (generators).
 * Fails with ZKGPU_EINVALID_POINT if any encoding is invalid. */
int zkgpu_pointset_create(zkgpu_ctx *ctx, const uint8_t *points, size_t n, zkgpu_pointset **out);
void zkgpu_pointset_destroy(zkgpu_pointset *ps);
size_t zkgpu_pointset_size(const zkgpu_pointset *ps);

/* Build fixed-base window tables for the set: for every window position t and
 * point j the affine multiples d * 2^(w t) * P_j, d = 1 .. 2^(w-1)
 * ((255/w + 1) * n * 2^(w-1) packed rows of 96 B, one per 128-byte line; w = 16, n = 514: 34.5 GB, n = 1026: 68.9 GB;
 * w = 12, n = 514: 2.2 GB).  With tables present zkgpu_verify_batch_ps* and every whole-proof
 * entry point sum static terms straight out of them: one mixed addition per term and window,
 * no doublings, no sorting.  One-time cost: ~0.2 s at 16 bits and 514 points (50 MB per point),
 * milliseconds at 8-12 bits; results are identical with or without tables.
 * 2 <= window_bits <= 16, or 0: the library chooses the KNEE -- among the widths whose tables take no more than a quarter
 * of the device's memory and whose construction (the tables + 1.7x scratch) fits in 60 % of what is free at the time of the
 * call, the narrowest whose additions per generator term stay within 19/16 of the widest feasible width's: 14 bits for 514
 * or 1026 points on an MI355X (10.2 GB / 20.4 GB at the 128-byte row stride instead of 34.5 / 68.9 GB at 16 bits, for
 * 0.5 - 2.5 % of verifier throughput; bench.py, setup.table_bits_sweep).  zkgpu_choose_table_bits answers the same question
 * without building; zkgpu_pointset_table_bits: the width in use, 0 without tables.  A PROVER gains from the widest tables
 * (16 bits: +15 %): pass the width explicitly. */
int zkgpu_pointset_build_tables(zkgpu_ctx *ctx, zkgpu_pointset *ps, int window_bits);
int zkgpu_choose_table_bits(zkgpu_ctx *ctx, size_t n_points);
int zkgpu_pointset_table_bits(const zkgpu_pointset *ps);
size_t zkgpu_pointset_table_bytes(const zkgpu_pointset *ps);

/* Values of `batch` multiscalar multiplications over a resident set with tables:
 * out[32 m] = ENCODE(sum_k scalars[k] * ps[index[k]]) for k in [offsets[m], offsets[m+1]);
 * index == NULL: term k of row m uses point k - offsets[m].  The prover-side primitive
 * (replaces: RistrettoPoint::multiscalar_mul over BulletproofGens / PedersenGens in
 * bulletproofs' r1cs::Prover -- A_I, A_O, S, T_i -- and in InnerProductProof::create when the
 * folded generators are kept as coefficient vectors over the original ones). */
int zkgpu_msm_ps_batch(zkgpu_ctx *ctx, const zkgpu_pointset *ps, size_t batch, const uint8_t *scalars,
                       const uint32_t *index, const uint64_t *offsets, uint8_t *out);

/* Proves `batch` ZkVM cloak statements of one shape (n_in inputs, n_out outputs) over the generator
 * set ps = [B, B_blinding, G_0..G_{cap-1}, H_0..H_{cap-1}] (tables built).  The provers run in
 * lockstep on `host_threads` host threads; each phase of the whole batch is one zkgpu_msm_ps_batch.
 * quantities: batch x (n_in + n_out) u64 (inputs first); flavors: 32 B per value; seeds: 32 B per
 * statement (commitment blindings and the TranscriptRng's external randomness derive from it).
 * Out: commitments batch x 64 (n_in + n_out) B (quantity, flavor per value); proofs batch x
 * proof_stride B, *proof_len bytes of each used (R1CSProof wire format, 1 + 32 (16 + 2k)).
 * (Replaces: spacesuit::cloak + bulletproofs r1cs::Prover::prove per transaction.) */
int zkgpu_cloak_prove_batch(zkgpu_ctx *ctx, const zkgpu_pointset *ps, size_t gens_capacity, size_t batch,
                            uint32_t n_in, uint32_t n_out, const uint64_t *quantities, const uint8_t *flavors,
                            const uint8_t *seeds, int host_threads, uint8_t *commitments, uint8_t *proofs,
                            size_t proof_stride, size_t *proof_len);

/* As zkgpu_verify_batch, with each check made of two CSR rows: "dynamic" terms
 * carrying their own compressed points (proof points, commitments) and "static"
 * terms that name a point of `ps` by index (generators).  static_index may be
 * NULL, meaning the j-th static term of a check uses point j of the set.
 * All pointers are host pointers; *_dev takes device pointers for everything
 * except accept_bitmap. */
int zkgpu_verify_batch_ps(zkgpu_ctx *ctx, const zkgpu_pointset *ps, size_t batch,
                          const uint8_t *dyn_scalars, const uint8_t *dyn_points,
                          const uint64_t *dyn_offsets,
                          const uint8_t *static_scalars, const uint32_t *static_index,
                          const uint64_t *static_offsets, uint8_t *accept_bitmap);
int zkgpu_verify_batch_ps_dev(zkgpu_ctx *ctx, const zkgpu_pointset *ps, size_t batch,
                              const void *d_dyn_scalars, const void *d_dyn_points,
                              const void *d_dyn_offsets, size_t n_dyn,
                              const void *d_static_scalars, const void *d_static_index,
                              const void *d_static_offsets, size_t n_static,
                              uint8_t *accept_bitmap);

/* Proof bytes in, accept bits out: the whole of bulletproofs::r1cs::Verifier::verify for
 * a batch of ZkVM `cloak` statements (what Tx::verify spends its time in once the VM has
 * run).  Statement i: n_in[i] inputs and n_out[i] outputs, each value a pair of Pedersen
 * commitments (quantity, flavor) -- commitments holds 64 * (n_in[i] + n_out[i]) bytes per
 * statement, back to back; proofs / proof_offsets are a CSR of R1CSProof encodings.
 * The transcript replay, constraint flattening and verification scalars run on
 * `host_threads` host threads (0 = as many as the process may keep busy: affinity mask and control-group CPU quota), the multiscalar multiplications in ONE device
 * call.  `ps` must hold [B, B_blinding, G_0..G_{cap-1}, H_0..H_{cap-1}] (zkgpu_pedersen_gens
 * + zkgpu_bulletproof_gens with gens_capacity = cap).  r_bytes: 64 uniform bytes per
 * statement for the verifier's random weight r, or NULL to draw them from the OS.
 * A malformed proof (wrong length, non-canonical scalar, identity where the reference's
 * validate_and_append_point forbids it, more multipliers than generators) clears its bit,
 * exactly where the reference returns Err. */
int zkgpu_cloak_verify_batch(zkgpu_ctx *ctx, const zkgpu_pointset *ps, size_t gens_capacity, size_t batch,
                             const uint32_t *n_in, const uint32_t *n_out, const uint8_t *commitments,
                             const uint8_t *proofs, const uint64_t *proof_offsets, const uint8_t *r_bytes,
                             uint8_t *accept_bitmap, int host_threads);

/* The same with the host half moved onto the device.  A plan compiles the constraint system
 * of one statement shape (n_in, n_out) once; zkgpu_cloak_verify_batch_gpu then replays the
 * Merlin transcripts (one lane per transaction), rebuilds every scalar of the verification
 * equation (one workgroup per transaction) and evaluates it, all on the GPU: the only PCIe
 * traffic is commitments + proof bytes + 64 bytes of verifier randomness per transaction.
 * Every statement of the batch has the plan's shape and `proof_len` bytes of proof
 * (fixed stride): the two-phase wire format (version byte 1, 16 + 2k elements) or, for a whole batch, the one-phase
 * one (version byte 0, A_I2 A_O2 S2 left out: 13 + 2k elements -- what upstream's R1CSProof::to_bytes writes for a
 * statement without a second phase; read as the identity for the three points).  A version byte that disagrees with the
 * length rejects that proof.  Blocks of mixed shapes (zkgpu_verifier_*) may mix the two forms freely.
 * Verdicts are identical to zkgpu_cloak_verify_batch. */
typedef struct zkgpu_cloak_plan zkgpu_cloak_plan;
int zkgpu_cloak_plan_create(zkgpu_ctx *ctx, uint32_t n_in, uint32_t n_out, size_t gens_capacity,
                            zkgpu_cloak_plan **out);
void zkgpu_cloak_plan_destroy(zkgpu_cloak_plan *plan);
int zkgpu_cloak_plan_info(const zkgpu_cloak_plan *plan, uint32_t *multipliers, uint32_t *padded_n,
                          uint32_t *constraints, uint32_t *terms, uint32_t *proof_len);
int zkgpu_cloak_verify_batch_gpu(zkgpu_ctx *ctx, const zkgpu_pointset *ps, zkgpu_cloak_plan *plan, size_t batch,
                                 const uint8_t *commitments, const uint8_t *proofs, size_t proof_len,
                                 const uint8_t *r_bytes, uint8_t *accept_bitmap);

/* ---- any constraint system, described as data ------------------------------
 * What lets Tx::verify use the device path for statements that are not a pure cloak: the caller (the Rust
 * host running the VM) traces its gadgets ONCE per statement shape with symbolic coefficients and hands over
 * the constraint system as arrays; zkgpu_cloak_plan_create is exactly this, with the library tracing the
 * cloak gadget itself.  Replaces: bulletproofs::r1cs::Verifier's constraint collection (constrain / multiply /
 * allocate_multiplier / specify_randomized_constraints + challenge_scalar) for the purposes of verification.
 *   - n_commitments m: the statement commits V_0 .. V_{m-1} (32 bytes each, in this order, per statement);
 *   - multipliers [0, n_multipliers_phase1) are allocated before the randomized constraints, the rest by them;
 *   - challenge_labels: Merlin labels of the challenge scalars the randomized constraints draw, in drawing order
 *     (none, and no second-phase multipliers: a single-phase statement, dom-sep "r1cs-1phase");
 *   - constraint q (in the order the constraint system emitted them -- the flattening weighs it z^(q+1)) is
 *         sum over its terms of  coefficient * variable  = 0
 *     with variable = (kind, index): 0 committed V_index, 1 / 2 / 3 left / right / output of multiplier index,
 *     4 the constant one; coefficient = c * challenge[term_challenge]^term_power, c a canonical 32-byte scalar,
 *     term_challenge = -1 for a plain constant.  (A coefficient involving two different challenges is not
 *     representable; no ZkVM gadget needs one.)
 * The resulting plan is a zkgpu_cloak_plan in all but name: every whole-proof entry point takes it
 * (zkgpu_r1cs_verify_* are the same functions under generic names; commitments = m x 32 bytes per statement;
 * proof_len = 1 + 32 (16 + 2 lg padded multipliers)).  zkgpu_r1cs_verify_batch is the host-prepared form
 * (transcript replay and scalars on host threads, as zkgpu_cloak_verify_batch). */
typedef struct zkgpu_r1cs_desc {
  const char *transcript_label;          /* Transcript::new(label), e.g. "ZkVM.r1cs" */
  uint32_t n_commitments;
  uint32_t n_multipliers_phase1;
  uint32_t n_multipliers;
  uint32_t n_challenges;
  const char *const *challenge_labels;
  uint32_t n_constraints;
  const uint64_t *term_offsets;          /* n_constraints + 1 */
  const uint8_t *term_var_kind;
  const uint32_t *term_var_index;
  const uint8_t *term_coeff;             /* 32 bytes per term */
  const int32_t *term_challenge;
  const uint32_t *term_power;
} zkgpu_r1cs_desc;
typedef zkgpu_cloak_plan zkgpu_r1cs_plan;
int zkgpu_r1cs_plan_create(zkgpu_ctx *ctx, const zkgpu_r1cs_desc *desc, size_t gens_capacity, zkgpu_r1cs_plan **out);
void zkgpu_r1cs_plan_destroy(zkgpu_r1cs_plan *plan);
int zkgpu_r1cs_verify_batch_gpu(zkgpu_ctx *ctx, const zkgpu_pointset *ps, zkgpu_r1cs_plan *plan, size_t batch,
                                const uint8_t *commitments, const uint8_t *proofs, size_t proof_len,
                                const uint8_t *r_bytes, uint8_t *accept_bitmap);
int zkgpu_r1cs_verify_submit(zkgpu_ctx *ctx, const zkgpu_pointset *ps, zkgpu_r1cs_plan *plan, size_t batch,
                             const uint8_t *commitments, const uint8_t *proofs, size_t proof_len, const uint8_t *r_bytes);
int zkgpu_r1cs_verify_submit_dev(zkgpu_ctx *ctx, const zkgpu_pointset *ps, zkgpu_r1cs_plan *plan, size_t batch,
                                 const void *d_commitments, const void *d_proofs, size_t proof_len, const void *d_r);
int zkgpu_r1cs_verify_batch(zkgpu_ctx *ctx, const zkgpu_pointset *ps, const zkgpu_r1cs_desc *desc, size_t gens_capacity,
                            size_t batch, const uint8_t *commitments, const uint8_t *proofs, size_t proof_len,
                            const uint8_t *r_bytes, uint8_t *accept_bitmap, int host_threads);

/* The prover for a described constraint system (BASELINE.json configs[4]: "R1CS proving -- Pedersen vector commits +
 * IPA -- for a 1024-constraint program"; replaces bulletproofs r1cs::Prover::{commit, constraint collection, prove}).
 * Beyond the description the prover needs the witness:
 *   values      batch x m x 32: the committed scalars (canonical); blindings: the same shape, or NULL to derive them
 *               from the statement's seed (SHAKE256(seed || "blinding" || LE64(j)) reduced mod l);
 *   mult_def    2 x n_multipliers: for multiplier i the indices of the two constraints that DEFINE its left and right
 *               input (`left - l_i = 0`, `right - r_i = 0`, as a multiply() emits them), or 0xffffffff twice when its
 *               assignment is given; NULL: every assignment is given.  Defined multipliers are evaluated in index
 *               order (second-phase ones after the challenges are drawn);
 *   given       batch x n_given x 64: (left, right) of the given multipliers, in index order;
 *   seeds       batch x 32: external randomness of the TranscriptRng (and the blindings when derived).
 * Out: commitments batch x 32 m, proofs batch x proof_stride (*proof_len bytes of each used).  The provers run in
 * lockstep on host_threads threads; every phase of the whole batch is one multiscalar-multiplication call on the
 * tables of `ps` = [B, B_blinding, G.., H..]. */
int zkgpu_r1cs_prove_batch(zkgpu_ctx *ctx, const zkgpu_pointset *ps, const zkgpu_r1cs_desc *desc, const uint32_t *mult_def,
                           size_t gens_capacity, size_t batch, const uint8_t *values, const uint8_t *blindings,
                           const uint8_t *given, size_t n_given, const uint8_t *seeds, int host_threads,
                           uint8_t *commitments, uint8_t *proofs, size_t proof_stride, size_t *proof_len);

/* Both provers run the whole proof on the device -- Merlin transcript, TranscriptRng, witness assignment from the described
 * system, constraint flattening, polynomial and inner-product algebra in kernels (one workgroup per proof), the multiscalar
 * multiplications on the tables in between, nothing of a proof in the making on the host; a call of 1024 statements or more
 * is cut into slices that are in flight together on streams of their own (the phases of one slice in the gaps of another's
 * multiplications).  The round-1 arrangement (host threads in lockstep) remains behind a hook (zkgpu_hooks.h). */

/* The same with commitments, proofs and verifier randomness already resident in HBM
 * (device pointers; this is what bench.py times as one step). */
int zkgpu_cloak_verify_batch_gpu_dev(zkgpu_ctx *ctx, const zkgpu_pointset *ps, zkgpu_cloak_plan *plan, size_t batch,
                                     const void *d_commitments, const void *d_proofs, size_t proof_len,
                                     const void *d_r, uint8_t *accept_bitmap);

/* ---- several batches in flight --------------------------------------------------------------
 * zkgpu_ctx_fork: a context with its own workspace and its own stream for the latency-bound
 * kernels (transcript replay, Horner tails) that SHARES the parent's two streams for the
 * chip-filling kernels, so that those run first-in first-out across the batches in flight.  The
 * parent must outlive its forks.  A fork is a full context: every entry point accepts it.
 *
 * zkgpu_cloak_verify_submit (host buffers: staged through pinned memory, free for reuse on return) /
 * zkgpu_cloak_verify_submit_dev / zkgpu_verify_batch_ps_submit_dev: as the calls of the same name
 * without `submit`, but return once the work is queued; at most one batch may be pending per
 * context.  zkgpu_verify_wait blocks until that batch is done and writes its accept bitmap
 * ((batch + 7) / 8 bytes; zeroed on any error: fail-closed).
 * (Replaces: a Rust caller running `Verifier::verify` for several blocks on a thread pool.) */
int zkgpu_ctx_fork(zkgpu_ctx* parent, zkgpu_ctx** out);

/* Group checks for whole proofs (zkgpu_cloak_verify_batch_gpu*, *_submit_dev): the verification
 * equations of 16 transactions are added up under independent random weights (the square of
 * each transaction's verifier randomness r) and checked by ONE multiscalar multiplication, in which
 * the generator terms of the whole group collapse into one set of scalars.  The accept bitmap is the one of
 * per-transaction verification (a bad transaction passes only with the ~2^-250 probability it already has against r).
 * (Same device as upstream bulletproofs' batched range-proof verification; applies to
 * r1cs::Verifier::verify of many proofs over one BulletproofGens.)
 * A group that fails is resolved with ONE more multiscalar multiplication when a single transaction is to
 * blame: with S1 = sum E_t and S2 = sum i_t E_t over the group (i_t = position), the culprit b satisfies
 * S2 = i_b S1; it alone is then checked on its own and the others are accepted iff S1 - E_b is the identity
 * (the group check restricted to them).  Groups with several bad transactions are re-checked one by one;
 * transactions known to be bad before the sums are formed (undecodable point, malformed proof) are
 * left out of their group.  Should a located transaction ever fail to account for its group (~2^-248),
 * zkgpu_verify_wait re-runs the batch ungrouped.  With *_dev inputs the caller's device buffers must
 * stay valid until zkgpu_verify_wait returns.  Below 2048 transactions per batch a failed group is simply re-checked
 * transaction by transaction (the two extra dependent stages cost more latency than the saved work is worth).
 * The transcripts are replayed one WAVEFRONT per transaction (Keccak-f with the state spread over the lanes,
 * keccak_coop.hpp) for batches of up to 1536 transactions and one LANE per transaction beyond; the Horner chains over the
 * windows run per GROUP while the batch the context family finished last had no failed group, else per transaction.
 * None of these choices changes a verdict; the alternatives every sweep has rejected, the group size, and the
 * device-side unit-test entry points are hooks (zkgpu_hooks.h), not exports. */

/* Plain device memory for callers without a HIP binding of their own: what the *_dev entry points
 * take.  zkgpu_upload is a blocking host-to-device copy. */
int zkgpu_malloc(zkgpu_ctx* ctx, size_t bytes, void** out);
int zkgpu_free(zkgpu_ctx* ctx, void* d_ptr);
int zkgpu_upload(zkgpu_ctx* ctx, void* d_dst, const void* src, size_t bytes);
int zkgpu_cloak_verify_submit(zkgpu_ctx* ctx, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch,
                              const uint8_t* commitments, const uint8_t* proofs, size_t proof_len,
                              const uint8_t* r_bytes);
int zkgpu_cloak_verify_submit_dev(zkgpu_ctx* ctx, const zkgpu_pointset* ps, zkgpu_cloak_plan* plan, size_t batch,
                                  const void* d_commitments, const void* d_proofs, size_t proof_len, const void* d_r);
int zkgpu_verify_batch_ps_submit_dev(zkgpu_ctx* ctx, const zkgpu_pointset* ps, size_t batch, const void* d_dyn_scalars,
                                     const void* d_dyn_points, const void* d_dyn_offsets, size_t n_dyn,
                                     const void* d_static_scalars, const void* d_static_index,
                                     const void* d_static_offsets, size_t n_static);
int zkgpu_verify_wait(zkgpu_ctx* ctx, uint8_t* accept_bitmap);

/* The host half of zkgpu_cloak_verify_batch alone: proof bytes -> the CSR of multiscalar
 * multiplication terms that zkgpu_verify_batch_ps* consumes (no device involved).
 * Statement i needs 11 + 2 (n_in+n_out) + 2k dynamic and 2 + 2 * 2^k static terms (k = lg of
 * its padded multiplier count); *_capacity are in terms.  wellformed[i] = 0 marks a proof
 * the reference would reject before the multiplication (its rows are left empty). */
int zkgpu_cloak_prepare_batch(size_t gens_capacity, size_t batch, const uint32_t *n_in, const uint32_t *n_out,
                              const uint8_t *commitments, const uint8_t *proofs, const uint64_t *proof_offsets,
                              const uint8_t *r_bytes, int host_threads, uint8_t *dyn_scalars, uint8_t *dyn_points,
                              uint64_t *dyn_offsets, size_t dyn_capacity, uint8_t *static_scalars,
                              uint32_t *static_index, uint64_t *static_offsets, size_t static_capacity,
                              uint8_t *wellformed);

/* Values instead of identity tests: out[32*i] = compress(sum of row i), or 32
 * zero bytes with bit i of ok_bitmap cleared when a point of row i is invalid
 * (RistrettoPoint::optional_multiscalar_mul returning None). Host pointers. */
int zkgpu_msm_batch(zkgpu_ctx *ctx, const uint8_t *scalars, const uint8_t *points,
                    const uint64_t *offsets, size_t batch, uint8_t *out, uint8_t *ok_bitmap);

/* RistrettoPoint::from_uniform_bytes for n 64-byte inputs (RFC 9496 sec 4.3.4),
 * out = n compressed points. */
int zkgpu_hash_to_points(zkgpu_ctx *ctx, const uint8_t *uniform, size_t n, uint8_t *out);

/* bulletproofs::PedersenGens::default() and BulletproofGens::new(capacity, ..).share(party),
 * as compressed points (feed them to zkgpu_pointset_create). */
int zkgpu_pedersen_gens(zkgpu_ctx *ctx, uint8_t B[32], uint8_t B_blinding[32]);
int zkgpu_bulletproof_gens(zkgpu_ctx *ctx, size_t capacity, uint32_t party, uint8_t *G, uint8_t *H);

/* Decode-only helper (CompressedRistretto::decompress validity): ok[i] = 1/0. */
int zkgpu_decode_check(zkgpu_ctx *ctx, const uint8_t *points, size_t n, uint8_t *ok);

/* ---- whole blocks of transactions of mixed shapes (BASELINE configs[3]) -----------------------
 * zkgpu_verifier mirrors the reference's `Verifier` for a block: it owns up to `batches_in_flight`
 * contexts (the given one and forks of it; 0 = default 6, at most 10.  Every context brings a stream of its own for
 * its latency-bound kernels; a lane whose stream does not really run beside the other lanes' -- the device co-schedules
 * fewer hardware queues than the runtime hands out -- is probed for at creation and NOT kept, so asking for more lanes
 * than the device serves costs nothing: zkgpu_verifier_queue_info reports lanes asked / kept / dropped; DESIGN.md
 * sec 5.1) and one device plan per statement shape, created on first use.  zkgpu_verifier_verify takes the block as Tx::verify sees it -- per
 * transaction (n_in, n_out), its 64 (n_in + n_out) commitment bytes back to back, its R1CSProof bytes
 * (CSR) and 64 bytes of verifier randomness (NULL: getrandom(2)) -- groups the transactions by shape,
 * verifies the groups as uniform batches of at most `chunk` transactions (default 2048) kept in flight
 * on its contexts, and writes one accept bit per transaction in block order.  A transaction whose
 * shape the generator set cannot serve (more multipliers than generators, no values, > 64 inputs or
 * outputs) or whose proof has the wrong length for its shape is rejected on its own, as in the
 * reference (InvalidGeneratorsLength / malformed proof); it never fails the block.
 * zkgpu_txblock keeps a block resident in HBM, grouped by shape (what bench.py --config 4 times);
 * zkgpu_verifier_verify = create + verify_block + destroy.  One call at a time per verifier. */
typedef struct zkgpu_verifier zkgpu_verifier;
typedef struct zkgpu_txblock zkgpu_txblock;
int zkgpu_verifier_create(zkgpu_ctx *ctx, const zkgpu_pointset *ps, size_t gens_capacity, int batches_in_flight,
                          zkgpu_verifier **out);
void zkgpu_verifier_destroy(zkgpu_verifier *v);
int zkgpu_verifier_set_chunk(zkgpu_verifier *v, size_t transactions);
int zkgpu_verifier_lanes(const zkgpu_verifier *v);
/* Hardware queues.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES queues (default 4; read ONCE, when the runtime
 * starts: the host exports what zkgpu_runtime_hint recommends BEFORE its first HIP call; the library itself never sets it).  Measured on MI355X (profiles/r03_hw_queues.txt): on 4 - 8 queues every
 * pair of streams still overlaps, but batches in flight that wait for each other's events take turns (a mixed block: 1.3
 * instead of 2.0 M tx/s); from 12 queues on that is gone, but the device runs fewer queues side by side than the runtime
 * hands out, and a lane whose light stream lands on a queue that is not co-scheduled with another lane's alternates with
 * it (the 3x - 10x cliff of round 2 at 8 and 10 lanes).  The library therefore does not trust the variable, it PROBES what
 * it got: every lane of a new verifier is checked against the lanes kept so far (two spinning wavefronts, ~0.2 ms per
 * pair, idle device assumed) and dropped when it would serialise with one of them -- fewer lanes, every one of them real;
 * verdicts are the same with any number of lanes.
 * zkgpu_verifier_queue_info: out[0] lanes in use, out[1] lanes asked for, out[2] lanes dropped, out[3] 1 when the
 * process's HIP runtime had started before GPU_MAX_HW_QUEUES was set (zkgpu_verifier_last_error names the cause).
 * zkgpu_ctx_queue_info: out[0] whether the context's two pipeline streams run side by side (1 / 0 / -1 probe failed;
 * zkgpu_init asks for another stream up to six times when they do not), out[1] GPU_MAX_HW_QUEUES as the process sees it
 * (0 = unset), out[2] the same late-start flag. */
int zkgpu_verifier_queue_info(const zkgpu_verifier *v, int out[4]);
int zkgpu_ctx_queue_info(zkgpu_ctx *ctx, int out[3]);
const char *zkgpu_verifier_last_error(const zkgpu_verifier *v);
int zkgpu_verifier_verify(zkgpu_verifier *v, size_t batch, const uint32_t *n_in, const uint32_t *n_out,
                          const uint8_t *commitments, const uint8_t *proofs, const uint64_t *proof_offsets,
                          const uint8_t *r_bytes, uint8_t *accept_bitmap);
int zkgpu_txblock_create(zkgpu_verifier *v, size_t batch, const uint32_t *n_in, const uint32_t *n_out,
                         const uint8_t *commitments, const uint8_t *proofs, const uint64_t *proof_offsets,
                         const uint8_t *r_bytes, zkgpu_txblock **out);
void zkgpu_txblock_destroy(zkgpu_txblock *block);
size_t zkgpu_txblock_size(const zkgpu_txblock *block);
size_t zkgpu_txblock_shapes(const zkgpu_txblock *block);
int zkgpu_verifier_verify_block(zkgpu_verifier *v, const zkgpu_txblock *block, uint8_t *accept_bitmap);
/* The same in two halves: zkgpu_verifier_block_start queues every batch of the block on the verifier's contexts and
 * returns a run id at once; zkgpu_verifier_block_finish waits for that run and writes the block's accept bitmap.  A
 * node that verifies a stream of blocks starts block k+1 before it finishes block k, so the chip does not drain between
 * blocks (bench.py --config 4 keeps two in flight).  When every context is busy, starting waits for the oldest batch in
 * flight first; its verdicts are kept with its run.  The block must stay alive until its run has been finished; runs are
 * finished in any order; an unknown or already finished run id: ZKGPU_EINVAL. */
int zkgpu_verifier_block_start(zkgpu_verifier *v, const zkgpu_txblock *block, uint64_t *run_id);
int zkgpu_verifier_block_finish(zkgpu_verifier *v, uint64_t run_id, uint8_t *accept_bitmap);
/* Tickets: many small batches in flight, few large device batches.  zkgpu_verifier_submit_dev QUEUES one uniform batch
 * (inputs resident in HBM, as zkgpu_cloak_verify_submit_dev) and returns a ticket at once; queued batches of one shape
 * are merged into device batches of about `merge` transactions (default 10 240 for tickets, 4096 for the batches of
 * blocks; zkgpu_verifier_set_merge sets both), copied side by side into the workspace of one of the verifier's contexts
 * (device to device) and launched as ONE batch as soon as the target is reached and a context is free -- or when a ticket
 * among them is waited for.  What is queued of one shape when a batch leaves is cut into round(queued / merge) batches of
 * EQUAL size, so a burst of 16, 20, 24 or 37 tickets of 1024 leaves as 2 x 8, 2 x 10, 2 x 12, 10 + 9 + 9 + 9: no run pays
 * for a short straggler batch (a device batch may therefore hold up to 1.5 x merge transactions).  zkgpu_verifier_wait blocks until that
 * ticket's batch is done and writes ITS accept bitmap (ceil(batch / 8) bytes; all zero on any error).  Verdicts are
 * those of separate batches; what changes is when a batch starts and how well it fills the chip: every kernel of a
 * 1024-transaction batch is a single round of workgroups, four merged batches run ~1.5x faster per transaction.
 * This is what bench.py times for the 1024-transaction batches of BASELINE.json configs[1].  Inputs must stay valid
 * until their ticket has been waited for; do not run zkgpu_verifier_verify* calls concurrently with tickets in flight
 * (they first finish them).  (A runtime policy -- dynamic batching -- with no counterpart in the reference.) */
int zkgpu_verifier_set_merge(zkgpu_verifier *v, size_t transactions);
/* Sizes every lane's workspace for device batches of `transactions` statements of the shape (n_in, n_out) beforehand, so that
 * no lane allocates -- hipMalloc synchronises the device -- when it first meets a batch that large: for verifiers whose
 * batches vary in shape and size (blocks of mixed shapes; since round 4 a block's batches are merged with those of the other
 * blocks in flight, up to the merge target).  Optional: without it workspaces grow on demand and are never shrunk. */
int zkgpu_verifier_reserve(zkgpu_verifier *v, uint32_t n_in, uint32_t n_out, size_t transactions);
int zkgpu_verifier_submit_dev(zkgpu_verifier *v, uint32_t n_in, uint32_t n_out, size_t batch, const void *d_commitments,
                              const void *d_proofs, size_t proof_len, const void *d_r, uint64_t *ticket);
/* the same for `count` batches of one shape and size at once (arrays of count device pointers; tickets[count]) */
int zkgpu_verifier_submit_many_dev(zkgpu_verifier *v, uint32_t n_in, uint32_t n_out, size_t count, size_t batch_each,
                                   const void *const *d_commitments, const void *const *d_proofs, size_t proof_len,
                                   const void *const *d_r, uint64_t *tickets);
/* Tickets from HOST memory -- what a caller of Tx::verify has: bytes in its own memory (upstream: the `R1CSProof` and the
 * commitments of `Tx`, zkvm `Verifier::verify_tx`; no file:line exists under /root/reference).  Same queue, same merging, same
 * zkgpu_verifier_wait; the inputs are copied into pinned staging memory DURING the call (they may be reused or freed as soon
 * as it returns) side by side with the other tickets of the device batch being formed, and reach the device as three
 * copies per DEVICE batch on a copy stream of the verifier's own, beside the batches in flight -- no merge kernel, no copy
 * per ticket.  r_bytes: 64 bytes of verifier randomness per transaction, or NULL (the OS's, expanded with SHAKE256).
 * A ticket of a shape the generator set cannot serve, or with a proof length that is wrong for its shape, is answered
 * with an all-zero bitmap and ZKGPU_OK by zkgpu_verifier_wait, as the reference answers each such transaction with Err. */
int zkgpu_verifier_submit(zkgpu_verifier *v, uint32_t n_in, uint32_t n_out, size_t batch, const uint8_t *commitments,
                          const uint8_t *proofs, size_t proof_len, const uint8_t *r_bytes, uint64_t *ticket);
/* `count` batches of one shape and size at once (arrays of count host pointers; r_bytes or any r_bytes[i] may be NULL) */
int zkgpu_verifier_submit_many(zkgpu_verifier *v, uint32_t n_in, uint32_t n_out, size_t count, size_t batch_each,
                               const uint8_t *const *commitments, const uint8_t *const *proofs, size_t proof_len,
                               const uint8_t *const *r_bytes, uint64_t *tickets);
int zkgpu_verifier_wait(zkgpu_verifier *v, uint64_t ticket, uint8_t *accept_bitmap);

/* ---- serialized transactions (replaces Tx::verify / Verifier::verify_tx for the PAYMENT SUBSET) --
 * txs: the transactions back to back, tx_offsets[batch + 1].  Per transaction, on host threads: wire format
 *     version u64 | mintime_ms u64 | maxtime_ms u64 | n u32 program[n] | R 32 s 32 | n u32 R1CSProof[n]      (little endian)
 * and the VM over the instructions  push:n:x  drop  dup:k  roll:k  var  cloak:m:n  input  output:k  signtx  (one cloak per
 * transaction), giving the transaction ID (Merkle root of the transaction log), the keys the signature must cover and the
 * cloak's statement; then, for the whole batch on the device: the MuSig-aggregated keys, the Schnorr equations
 * s B = R + c X as multiscalar multiplications == identity, and the cloak proofs through zkgpu_verifier_verify.
 * accept bit i = all of it holds.  status (optional, batch bytes): 0 accepted, 1 rejected, 2 OUTSIDE THE SUBSET (another
 * instruction, another version, no or several cloaks): the caller's own VM must decide -- such a transaction is never
 * reported as invalid.  UNPINNED: format, opcodes and labels follow a recollection of the public ZkVM design notes
 * (DESIGN.md sec 4.5); nothing under /root/reference defines them.
 * EXPERIMENTAL, OPT-IN: because the format is unpinned the entry point is inert by default -- every transaction is
 * reported as OUTSIDE THE SUBSET (status 2, accept bit 0: "ask your own VM") until the caller names the format with
 * zkgpu_verifier_set_tx_format(v, ZKGPU_TXFORMAT_RECOLLECTED_V1).  Both outputs are fail-closed: status 0 is written
 * only beside an accept bit of 1, after every stage has passed; on any error every transaction inside the subset reads
 * "rejected" and the bitmap is zero.  The call holds the verifier for its whole length and first collects whatever its
 * lanes have in flight (tickets and blocks keep their verdicts).
 * Inside, the call is cut into chunks (equal parts of at most 8192 transactions, 4096 for calls longer than 16 384;
 * zkgpu_verifier_set_tx_chunk, 0 = automatic) that travel through the stages -- a first VM pass as far as the signature's
 * keys, the aggregated keys and the cloak proofs (staged through pinned memory, queued on the lanes) on the device at
 * once; the rest of the VM's hashing beside them; then the signature equations -- on a staging thread of the call's own
 * (its loops on the library's worker pool, `host_threads` wide) and the calling thread, which only talks to the device.
 * The verifier keeps what the VM leaves per transaction (700 bytes) between calls, for at most
 * zkgpu_verifier_set_tx_statements_kept transactions (default 131 072, i.e. 90 MB; what a longer call needs beyond
 * that is allocated for the call). */
#define ZKGPU_TXFORMAT_RECOLLECTED_V1 1
int zkgpu_verifier_set_tx_format(zkgpu_verifier *v, int format);
int zkgpu_verifier_set_tx_chunk(zkgpu_verifier *v, size_t transactions);
int zkgpu_verifier_set_tx_statements_kept(zkgpu_verifier *v, size_t transactions);
int zkgpu_tx_verify_batch(zkgpu_verifier *v, size_t batch, const uint8_t *txs, const uint64_t *tx_offsets, int host_threads,
                          uint8_t *accept_bitmap, uint8_t *status);
/* Calls in flight (upstream's Tx::verify is pure and callable from many threads: SURVEY.md sec 8(b)).  zkgpu_tx_verify_submit
 * queues a call and returns its id at once; an engine thread of the verifier runs everything that is queued as ONE merged
 * call (up to 16 384 transactions per round: calls that arrive while a round runs make up the next one -- dynamic batching,
 * as tickets do for proofs; two rounds in flight while GPU_MAX_HW_QUEUES is 8..19, the library's own setting, else one), and zkgpu_tx_verify_wait blocks until that call's round is done and writes ITS accept bitmap and
 * status bytes (status may be NULL).  txs and tx_offsets must stay valid until the call has been waited for; every id is
 * waited for exactly once.  Same verdicts, same opt-in format, same fail-closed rule as zkgpu_tx_verify_batch: a round that
 * fails gives every call in it the error and all-zero outputs.  Safe to call from many threads.  zkgpu_tx_verify_stats:
 * out[0] rounds run so far, out[1] calls they held. */
int zkgpu_tx_verify_submit(zkgpu_verifier *v, size_t batch, const uint8_t *txs, const uint64_t *tx_offsets, int host_threads,
                           uint64_t *call_id);
int zkgpu_tx_verify_wait(zkgpu_verifier *v, uint64_t call_id, uint8_t *accept_bitmap, uint8_t *status);
int zkgpu_tx_verify_stats(zkgpu_verifier *v, uint64_t out[2]);

/* ---- one process per GPU: sharding and the RCCL exchange ------------------
 * Transactions are independent, so a block is cut into `world` contiguous shards balanced by the
 * number of multiscalar-multiplication terms (zkgpu_cloak_msm_terms of each shape; zkgpu_shard_cuts
 * fills cuts[0..world]), rank r verifies [cuts[r], cuts[r+1]) on its own GPU, and the only exchange is
 * one ncclAllGather of the per-shard accept bitmaps (a status word + ceil(shard/8) bytes per rank:
 * latency-bound, xGMI bandwidth irrelevant).  zkgpu_comm wraps the communicator: rank 0 calls
 * zkgpu_comm_unique_id and ships the 128 bytes to the other processes by whatever channel the host
 * has; every process then calls zkgpu_comm_create on its own context.  RCCL is bound with dlopen at
 * first use (librccl.so.1): without it these calls return ZKGPU_ENOCOMM and nothing else is affected;
 * a world of one needs no RCCL (id == NULL).
 * zkgpu_comm_allgather_bitmap is fail-closed ACROSS ranks: if any rank passes a non-zero local_status
 * every rank gets an all-zero bitmap and an error (its own, or ZKGPU_EREMOTE) -- and no rank is left
 * waiting in the collective: a communicator owns fixed exchange buffers from its creation (a rank contributes at most
 * 1 MiB per call: a status word and the bitmap of 8 M transactions), so nothing that can fail on ONE rank stands between
 * a call and its collective; a local fault (the caller's status, a missing bitmap, a failed copy to the device) travels
 * through the gather as that rank's status word.  Arguments that are the same on every rank (the cuts) are checked
 * before it.  Framing: csrc/comm_frame.hpp.  zkgpu_verifier_verify_sharded = cuts + verify own shard + that gather,
 * with the whole block in host memory on every rank.  (The exchange step is exercised at world 2 .. 8 on one GPU through
 * the hook zkgpu_debug_comm_mock, zkgpu_hooks.h; a communicator keeps the function table it was created with.) */
#define ZKGPU_COMM_ID_BYTES 128
typedef struct zkgpu_comm zkgpu_comm;
uint64_t zkgpu_cloak_msm_terms(uint32_t n_in, uint32_t n_out);
int zkgpu_shard_cuts(size_t batch, const uint32_t *n_in, const uint32_t *n_out, int world, uint64_t *cuts);
int zkgpu_comm_unique_id(uint8_t id[ZKGPU_COMM_ID_BYTES]);
int zkgpu_comm_create(zkgpu_ctx *ctx, int rank, int world, const uint8_t id[ZKGPU_COMM_ID_BYTES], zkgpu_comm **out);
void zkgpu_comm_destroy(zkgpu_comm *comm);
int zkgpu_comm_rank(const zkgpu_comm *comm);
int zkgpu_comm_world(const zkgpu_comm *comm);
int zkgpu_comm_allgather(zkgpu_comm *comm, const uint8_t *local, size_t bytes, uint8_t *all);
int zkgpu_comm_allgather_bitmap(zkgpu_comm *comm, const uint64_t *cuts, const uint8_t *local_bitmap, int local_status,
                                uint8_t *whole_bitmap);
int zkgpu_verifier_verify_sharded(zkgpu_verifier *v, zkgpu_comm *comm, size_t batch, const uint32_t *n_in,
                                  const uint32_t *n_out, const uint8_t *commitments, const uint8_t *proofs,
                                  const uint64_t *proof_offsets, const uint8_t *r_bytes, uint8_t *accept_bitmap);

/* ---- hooks ---------------------------------------------------------------------------------------------------
 * Per-kernel timing, the tuning modes that the sweeps of DESIGN.md have settled, the device-side unit tests of the
 * arithmetic layers and the mocked collective are NOT exports of this library: they are declared in zkgpu_hooks.h and
 * reached by name through this one function, which answers only when ZKGPU_TEST_HOOKS=1 was in the environment when
 * the library was loaded (tests/conftest.py and bench.py set it; a deployed verifier never does) and returns NULL
 * otherwise -- an integrator cannot link against them, and `nm -D` shows none of them.
 * Environment variables read by the library, none of which changes a result:
 *   ZKGPU_TIMELINE=<file>     with profiling on, every launch as "ctx kernel start_ms end_ms"
 *   ZKGPU_PROVER_TIMING=1     the provers and zkgpu_tx_verify_batch print per-phase host / device times to stderr
 *   ZKGPU_PROVER_SLICES=n     slices of a prover call in flight together (default: 2 from 1024 statements on)
 * and GPU_MAX_HW_QUEUES (a HIP runtime variable, READ only: the host exports it on zkgpu_runtime_hint's advice before its
 * first HIP call; batches in flight need a hardware queue per context). */
const void *zkgpu_hook(const char *name);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
