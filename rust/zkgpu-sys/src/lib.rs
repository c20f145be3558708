//! Raw FFI declarations of `libzkgpu.so` (include/zkgpu.h, ABI version 3): the MI355X back end of the ZkVM /
//! Bulletproofs-R1CS verification path.  One declaration per exported function, in the header's order; the
//! header's comments are the documentation (conventions, ownership, fail-closed rules) and are not repeated here.
//!
//! Produced from the header by `tools/gen_rust_sys.py` and committed as source; `tests/test_rust_binding.py`
//! re-parses both files and fails on any mismatch of name, arity, integer width, pointer depth or constness.
//! NOT COMPILED in the build container (no rustc / cargo there): a maintainer runs `cargo check` first.
#![allow(non_camel_case_types, non_snake_case)]

use std::os::raw::{c_char, c_int, c_longlong, c_void};

pub const ZKGPU_OK: c_int = 0;
pub const ZKGPU_EINVAL: c_int = -1;
pub const ZKGPU_EINVALID_POINT: c_int = -2;
pub const ZKGPU_EHIP: c_int = -3;
pub const ZKGPU_ENOMEM: c_int = -4;
pub const ZKGPU_ENODEVICE: c_int = -5;
pub const ZKGPU_ENOCOMM: c_int = -6;
pub const ZKGPU_EREMOTE: c_int = -7;
pub const ZKGPU_WSECOND_VERIFIER: c_int = 1;
pub const ZKGPU_HINT_APPLY: c_int = 0;
pub const ZKGPU_HINT_PRESENT: c_int = 1;
pub const ZKGPU_HINT_LATE: c_int = 2;
pub const ZKGPU_TXFORMAT_RECOLLECTED_V1: c_int = 1;
pub const ZKGPU_COMM_ID_BYTES: usize = 128;

#[repr(C)]
pub struct zkgpu_cloak_plan {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkgpu_comm {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkgpu_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkgpu_pointset {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkgpu_txblock {
    _private: [u8; 0],
}
#[repr(C)]
pub struct zkgpu_verifier {
    _private: [u8; 0],
}
/// `typedef zkgpu_cloak_plan zkgpu_r1cs_plan;` -- one type under two names
pub type zkgpu_r1cs_plan = zkgpu_cloak_plan;

/// A constraint system handed over as data (see the header for the meaning of every array).
#[repr(C)]
pub struct zkgpu_r1cs_desc {
    pub transcript_label: *const c_char,
    pub n_commitments: u32,
    pub n_multipliers_phase1: u32,
    pub n_multipliers: u32,
    pub n_challenges: u32,
    pub challenge_labels: *const *const c_char,
    pub n_constraints: u32,
    pub term_offsets: *const u64,
    pub term_var_kind: *const u8,
    pub term_var_index: *const u32,
    pub term_coeff: *const u8,
    pub term_challenge: *const c_int,
    pub term_power: *const u32,
}

#[link(name = "zkgpu")]
extern "C" {
    pub fn zkgpu_abi_version() -> c_int;
    pub fn zkgpu_strerror(code: c_int) -> *const c_char;
    pub fn zkgpu_last_error(ctx: *const zkgpu_ctx) -> *const c_char;
    pub fn zkgpu_runtime_hint(buf: *mut c_char, cap: usize) -> c_int;
    pub fn zkgpu_init(device: c_int, out: *mut *mut zkgpu_ctx) -> c_int;
    pub fn zkgpu_destroy(ctx: *mut zkgpu_ctx);
    pub fn zkgpu_msm(
        ctx: *mut zkgpu_ctx, scalars: *const u8, points: *const u8, n: usize, out: *mut u8, bad_index: *mut usize,
    ) -> c_int;
    pub fn zkgpu_msm_dev(
        ctx: *mut zkgpu_ctx, d_scalars: *const c_void, d_points: *const c_void, n: usize, out: *mut u8,
        bad_index: *mut usize,
    ) -> c_int;
    pub fn zkgpu_verify_batch(
        ctx: *mut zkgpu_ctx, scalars: *const u8, points: *const u8, offsets: *const u64, batch: usize,
        accept_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_verify_batch_dev(
        ctx: *mut zkgpu_ctx, d_scalars: *const c_void, d_points: *const c_void, d_offsets: *const c_void,
        batch: usize, n_terms: usize, accept_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_pointset_create(
        ctx: *mut zkgpu_ctx, points: *const u8, n: usize, out: *mut *mut zkgpu_pointset,
    ) -> c_int;
    pub fn zkgpu_pointset_destroy(ps: *mut zkgpu_pointset);
    pub fn zkgpu_pointset_size(ps: *const zkgpu_pointset) -> usize;
    pub fn zkgpu_pointset_build_tables(ctx: *mut zkgpu_ctx, ps: *mut zkgpu_pointset, window_bits: c_int) -> c_int;
    pub fn zkgpu_choose_table_bits(ctx: *mut zkgpu_ctx, n_points: usize) -> c_int;
    pub fn zkgpu_pointset_table_bits(ps: *const zkgpu_pointset) -> c_int;
    pub fn zkgpu_pointset_table_bytes(ps: *const zkgpu_pointset) -> usize;
    pub fn zkgpu_msm_ps_batch(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, batch: usize, scalars: *const u8, index: *const u32,
        offsets: *const u64, out: *mut u8,
    ) -> c_int;
    pub fn zkgpu_cloak_prove_batch(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, gens_capacity: usize, batch: usize, n_in: u32, n_out: u32,
        quantities: *const u64, flavors: *const u8, seeds: *const u8, host_threads: c_int, commitments: *mut u8,
        proofs: *mut u8, proof_stride: usize, proof_len: *mut usize,
    ) -> c_int;
    pub fn zkgpu_verify_batch_ps(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, batch: usize, dyn_scalars: *const u8, dyn_points: *const u8,
        dyn_offsets: *const u64, static_scalars: *const u8, static_index: *const u32, static_offsets: *const u64,
        accept_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_verify_batch_ps_dev(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, batch: usize, d_dyn_scalars: *const c_void,
        d_dyn_points: *const c_void, d_dyn_offsets: *const c_void, n_dyn: usize, d_static_scalars: *const c_void,
        d_static_index: *const c_void, d_static_offsets: *const c_void, n_static: usize, accept_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_cloak_verify_batch(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, gens_capacity: usize, batch: usize, n_in: *const u32,
        n_out: *const u32, commitments: *const u8, proofs: *const u8, proof_offsets: *const u64, r_bytes: *const u8,
        accept_bitmap: *mut u8, host_threads: c_int,
    ) -> c_int;
    pub fn zkgpu_cloak_plan_create(
        ctx: *mut zkgpu_ctx, n_in: u32, n_out: u32, gens_capacity: usize, out: *mut *mut zkgpu_cloak_plan,
    ) -> c_int;
    pub fn zkgpu_cloak_plan_destroy(plan: *mut zkgpu_cloak_plan);
    pub fn zkgpu_cloak_plan_info(
        plan: *const zkgpu_cloak_plan, multipliers: *mut u32, padded_n: *mut u32, constraints: *mut u32,
        terms: *mut u32, proof_len: *mut u32,
    ) -> c_int;
    pub fn zkgpu_cloak_verify_batch_gpu(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, plan: *mut zkgpu_cloak_plan, batch: usize,
        commitments: *const u8, proofs: *const u8, proof_len: usize, r_bytes: *const u8, accept_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_r1cs_plan_create(
        ctx: *mut zkgpu_ctx, desc: *const zkgpu_r1cs_desc, gens_capacity: usize, out: *mut *mut zkgpu_cloak_plan,
    ) -> c_int;
    pub fn zkgpu_r1cs_plan_destroy(plan: *mut zkgpu_cloak_plan);
    pub fn zkgpu_r1cs_verify_batch_gpu(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, plan: *mut zkgpu_cloak_plan, batch: usize,
        commitments: *const u8, proofs: *const u8, proof_len: usize, r_bytes: *const u8, accept_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_r1cs_verify_submit(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, plan: *mut zkgpu_cloak_plan, batch: usize,
        commitments: *const u8, proofs: *const u8, proof_len: usize, r_bytes: *const u8,
    ) -> c_int;
    pub fn zkgpu_r1cs_verify_submit_dev(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, plan: *mut zkgpu_cloak_plan, batch: usize,
        d_commitments: *const c_void, d_proofs: *const c_void, proof_len: usize, d_r: *const c_void,
    ) -> c_int;
    pub fn zkgpu_r1cs_verify_batch(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, desc: *const zkgpu_r1cs_desc, gens_capacity: usize,
        batch: usize, commitments: *const u8, proofs: *const u8, proof_len: usize, r_bytes: *const u8,
        accept_bitmap: *mut u8, host_threads: c_int,
    ) -> c_int;
    pub fn zkgpu_r1cs_prove_batch(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, desc: *const zkgpu_r1cs_desc, mult_def: *const u32,
        gens_capacity: usize, batch: usize, values: *const u8, blindings: *const u8, given: *const u8, n_given: usize,
        seeds: *const u8, host_threads: c_int, commitments: *mut u8, proofs: *mut u8, proof_stride: usize,
        proof_len: *mut usize,
    ) -> c_int;
    pub fn zkgpu_cloak_verify_batch_gpu_dev(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, plan: *mut zkgpu_cloak_plan, batch: usize,
        d_commitments: *const c_void, d_proofs: *const c_void, proof_len: usize, d_r: *const c_void,
        accept_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_ctx_fork(parent: *mut zkgpu_ctx, out: *mut *mut zkgpu_ctx) -> c_int;
    pub fn zkgpu_malloc(ctx: *mut zkgpu_ctx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn zkgpu_free(ctx: *mut zkgpu_ctx, d_ptr: *mut c_void) -> c_int;
    pub fn zkgpu_upload(ctx: *mut zkgpu_ctx, d_dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn zkgpu_cloak_verify_submit(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, plan: *mut zkgpu_cloak_plan, batch: usize,
        commitments: *const u8, proofs: *const u8, proof_len: usize, r_bytes: *const u8,
    ) -> c_int;
    pub fn zkgpu_cloak_verify_submit_dev(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, plan: *mut zkgpu_cloak_plan, batch: usize,
        d_commitments: *const c_void, d_proofs: *const c_void, proof_len: usize, d_r: *const c_void,
    ) -> c_int;
    pub fn zkgpu_verify_batch_ps_submit_dev(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, batch: usize, d_dyn_scalars: *const c_void,
        d_dyn_points: *const c_void, d_dyn_offsets: *const c_void, n_dyn: usize, d_static_scalars: *const c_void,
        d_static_index: *const c_void, d_static_offsets: *const c_void, n_static: usize,
    ) -> c_int;
    pub fn zkgpu_verify_wait(ctx: *mut zkgpu_ctx, accept_bitmap: *mut u8) -> c_int;
    pub fn zkgpu_cloak_prepare_batch(
        gens_capacity: usize, batch: usize, n_in: *const u32, n_out: *const u32, commitments: *const u8,
        proofs: *const u8, proof_offsets: *const u64, r_bytes: *const u8, host_threads: c_int, dyn_scalars: *mut u8,
        dyn_points: *mut u8, dyn_offsets: *mut u64, dyn_capacity: usize, static_scalars: *mut u8,
        static_index: *mut u32, static_offsets: *mut u64, static_capacity: usize, wellformed: *mut u8,
    ) -> c_int;
    pub fn zkgpu_msm_batch(
        ctx: *mut zkgpu_ctx, scalars: *const u8, points: *const u8, offsets: *const u64, batch: usize, out: *mut u8,
        ok_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_hash_to_points(ctx: *mut zkgpu_ctx, uniform: *const u8, n: usize, out: *mut u8) -> c_int;
    pub fn zkgpu_pedersen_gens(ctx: *mut zkgpu_ctx, B: *mut u8, B_blinding: *mut u8) -> c_int;
    pub fn zkgpu_bulletproof_gens(ctx: *mut zkgpu_ctx, capacity: usize, party: u32, G: *mut u8, H: *mut u8) -> c_int;
    pub fn zkgpu_decode_check(ctx: *mut zkgpu_ctx, points: *const u8, n: usize, ok: *mut u8) -> c_int;
    pub fn zkgpu_verifier_create(
        ctx: *mut zkgpu_ctx, ps: *const zkgpu_pointset, gens_capacity: usize, batches_in_flight: c_int,
        out: *mut *mut zkgpu_verifier,
    ) -> c_int;
    pub fn zkgpu_verifier_destroy(v: *mut zkgpu_verifier);
    pub fn zkgpu_verifier_set_chunk(v: *mut zkgpu_verifier, transactions: usize) -> c_int;
    pub fn zkgpu_verifier_lanes(v: *const zkgpu_verifier) -> c_int;
    pub fn zkgpu_verifier_queue_info(v: *const zkgpu_verifier, out: *mut c_int) -> c_int;
    pub fn zkgpu_ctx_queue_info(ctx: *mut zkgpu_ctx, out: *mut c_int) -> c_int;
    pub fn zkgpu_verifier_last_error(v: *const zkgpu_verifier) -> *const c_char;
    pub fn zkgpu_verifier_verify(
        v: *mut zkgpu_verifier, batch: usize, n_in: *const u32, n_out: *const u32, commitments: *const u8,
        proofs: *const u8, proof_offsets: *const u64, r_bytes: *const u8, accept_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_txblock_create(
        v: *mut zkgpu_verifier, batch: usize, n_in: *const u32, n_out: *const u32, commitments: *const u8,
        proofs: *const u8, proof_offsets: *const u64, r_bytes: *const u8, out: *mut *mut zkgpu_txblock,
    ) -> c_int;
    pub fn zkgpu_txblock_destroy(block: *mut zkgpu_txblock);
    pub fn zkgpu_txblock_size(block: *const zkgpu_txblock) -> usize;
    pub fn zkgpu_txblock_shapes(block: *const zkgpu_txblock) -> usize;
    pub fn zkgpu_verifier_verify_block(
        v: *mut zkgpu_verifier, block: *const zkgpu_txblock, accept_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_verifier_block_start(v: *mut zkgpu_verifier, block: *const zkgpu_txblock, run_id: *mut u64) -> c_int;
    pub fn zkgpu_verifier_block_finish(v: *mut zkgpu_verifier, run_id: u64, accept_bitmap: *mut u8) -> c_int;
    pub fn zkgpu_verifier_set_merge(v: *mut zkgpu_verifier, transactions: usize) -> c_int;
    pub fn zkgpu_verifier_reserve(v: *mut zkgpu_verifier, n_in: u32, n_out: u32, transactions: usize) -> c_int;
    pub fn zkgpu_verifier_submit_dev(
        v: *mut zkgpu_verifier, n_in: u32, n_out: u32, batch: usize, d_commitments: *const c_void,
        d_proofs: *const c_void, proof_len: usize, d_r: *const c_void, ticket: *mut u64,
    ) -> c_int;
    pub fn zkgpu_verifier_submit_many_dev(
        v: *mut zkgpu_verifier, n_in: u32, n_out: u32, count: usize, batch_each: usize,
        d_commitments: *const *const c_void, d_proofs: *const *const c_void, proof_len: usize,
        d_r: *const *const c_void, tickets: *mut u64,
    ) -> c_int;
    pub fn zkgpu_verifier_submit(
        v: *mut zkgpu_verifier, n_in: u32, n_out: u32, batch: usize, commitments: *const u8, proofs: *const u8,
        proof_len: usize, r_bytes: *const u8, ticket: *mut u64,
    ) -> c_int;
    pub fn zkgpu_verifier_submit_many(
        v: *mut zkgpu_verifier, n_in: u32, n_out: u32, count: usize, batch_each: usize, commitments: *const *const u8,
        proofs: *const *const u8, proof_len: usize, r_bytes: *const *const u8, tickets: *mut u64,
    ) -> c_int;
    pub fn zkgpu_verifier_wait(v: *mut zkgpu_verifier, ticket: u64, accept_bitmap: *mut u8) -> c_int;
    pub fn zkgpu_verifier_set_tx_format(v: *mut zkgpu_verifier, format: c_int) -> c_int;
    pub fn zkgpu_verifier_set_tx_chunk(v: *mut zkgpu_verifier, transactions: usize) -> c_int;
    pub fn zkgpu_verifier_set_tx_statements_kept(v: *mut zkgpu_verifier, transactions: usize) -> c_int;
    pub fn zkgpu_tx_verify_batch(
        v: *mut zkgpu_verifier, batch: usize, txs: *const u8, tx_offsets: *const u64, host_threads: c_int,
        accept_bitmap: *mut u8, status: *mut u8,
    ) -> c_int;
    pub fn zkgpu_tx_verify_submit(
        v: *mut zkgpu_verifier, batch: usize, txs: *const u8, tx_offsets: *const u64, host_threads: c_int,
        call_id: *mut u64,
    ) -> c_int;
    pub fn zkgpu_tx_verify_wait(v: *mut zkgpu_verifier, call_id: u64, accept_bitmap: *mut u8, status: *mut u8) -> c_int;
    pub fn zkgpu_tx_verify_stats(v: *mut zkgpu_verifier, out: *mut u64) -> c_int;
    pub fn zkgpu_cloak_msm_terms(n_in: u32, n_out: u32) -> u64;
    pub fn zkgpu_shard_cuts(batch: usize, n_in: *const u32, n_out: *const u32, world: c_int, cuts: *mut u64) -> c_int;
    pub fn zkgpu_comm_unique_id(id: *mut u8) -> c_int;
    pub fn zkgpu_comm_create(
        ctx: *mut zkgpu_ctx, rank: c_int, world: c_int, id: *const u8, out: *mut *mut zkgpu_comm,
    ) -> c_int;
    pub fn zkgpu_comm_destroy(comm: *mut zkgpu_comm);
    pub fn zkgpu_comm_rank(comm: *const zkgpu_comm) -> c_int;
    pub fn zkgpu_comm_world(comm: *const zkgpu_comm) -> c_int;
    pub fn zkgpu_comm_allgather(comm: *mut zkgpu_comm, local: *const u8, bytes: usize, all: *mut u8) -> c_int;
    pub fn zkgpu_comm_allgather_bitmap(
        comm: *mut zkgpu_comm, cuts: *const u64, local_bitmap: *const u8, local_status: c_int, whole_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_verifier_verify_sharded(
        v: *mut zkgpu_verifier, comm: *mut zkgpu_comm, batch: usize, n_in: *const u32, n_out: *const u32,
        commitments: *const u8, proofs: *const u8, proof_offsets: *const u64, r_bytes: *const u8,
        accept_bitmap: *mut u8,
    ) -> c_int;
    pub fn zkgpu_hook(name: *const c_char) -> *const c_void;
}
