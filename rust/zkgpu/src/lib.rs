//! Safe layer over `zkgpu-sys`: the device side of `zkvm::Tx::verify` / `Verifier::verify_tx` and of
//! `bulletproofs::r1cs::Verifier::verify` on an AMD MI355X (upstream names as recalled in SURVEY.md sec 3; no file:line
//! exists under /root/reference, which holds no source).
//!
//! Everything here works on byte slices -- `Scalar::as_bytes()`, `CompressedRistretto::as_bytes()`,
//! `R1CSProof::to_bytes()`, `Tx::encode()` are already the layouts the C ABI takes -- so the crate depends on no upstream
//! crate and can be dropped under `zkvm` behind a cargo feature:
//!
//! ```ignore
//! // zkvm/src/verifier.rs, feature "gpu"
//! static GPU: once_cell::sync::Lazy<zkgpu::GpuVerifier> = Lazy::new(|| zkgpu::GpuVerifier::new(0, 256, 0).unwrap());
//! pub fn verify_block(txs: &[Tx], bp_gens: &BulletproofGens) -> Vec<Result<VerifiedTx, VMError>> {
//!     let stmts: Vec<_> = txs.iter().map(|tx| vm_run(tx)).collect();          // VM per transaction, as today
//!     let cloak: Vec<zkgpu::CloakStatement> = stmts.iter().map(|s| s.as_statement()).collect();
//!     let ok = GPU.verify_block(&cloak, None).unwrap_or_else(|_| vec![false; txs.len()]);   // fail-closed
//!     ...
//! }
//! ```
//!
//! NOT COMPILED in the build container (no rustc there).  `tests/test_rust_binding.py` checks every `sys::` call below
//! against the raw declarations (existence, number of arguments) and the raw declarations against `include/zkgpu.h`.
//!
//! Rules of the library that shape this API (include/zkgpu.h, INTEGRATION.md sec 3d''):
//! * fail-closed: any device error gives an `Err` and NO verdicts; a verdict `true` is only ever produced by a completed
//!   identity test on the device;
//! * one `GpuVerifier` per process and device; calls on it are serialised by the library (`&self` methods are safe to call
//!   from many threads: they queue);
//! * tickets (`submit` / `wait`) are how small batches reach the device's efficient batch size: they are merged into
//!   device batches of `merge` transactions.
#![allow(clippy::too_many_arguments)]

use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;

use zkgpu_sys as sys;

/// What a call into the library can fail with (`ZKGPU_E*` of the header).  `Hip` and `Comm` carry the library's text.
#[derive(Debug, Clone, PartialEq, Eq)]
pub enum Error {
    InvalidArgument(String),
    InvalidPoint,
    Hip(String),
    OutOfMemory,
    NoDevice,
    Comm(String),
    RemoteRank,
    AbiMismatch(i32),
    Other(i32),
}

impl std::fmt::Display for Error {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "{:?}", self)
    }
}
impl std::error::Error for Error {}

fn text(p: *const std::os::raw::c_char) -> String {
    if p.is_null() {
        String::new()
    } else {
        unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
    }
}

fn check(code: c_int, detail: impl FnOnce() -> String) -> Result<(), Error> {
    match code {
        sys::ZKGPU_OK => Ok(()),
        sys::ZKGPU_EINVAL => Err(Error::InvalidArgument(detail())),
        sys::ZKGPU_EINVALID_POINT => Err(Error::InvalidPoint),
        sys::ZKGPU_EHIP => Err(Error::Hip(detail())),
        sys::ZKGPU_ENOMEM => Err(Error::OutOfMemory),
        sys::ZKGPU_ENODEVICE => Err(Error::NoDevice),
        sys::ZKGPU_ENOCOMM => Err(Error::Comm(detail())),
        sys::ZKGPU_EREMOTE => Err(Error::RemoteRank),
        other => Err(Error::Other(other)),
    }
}

fn bits(bitmap: &[u8], n: usize) -> Vec<bool> {
    (0..n).map(|i| (bitmap[i / 8] >> (i % 8)) & 1 == 1).collect()
}

/// One cloak statement as the VM leaves it: `commitments` = 64 bytes per value (quantity, flavor), the `n_in` inputs
/// first; `proof` = `R1CSProof::to_bytes()`.
#[derive(Clone, Copy)]
pub struct CloakStatement<'a> {
    pub n_in: u32,
    pub n_out: u32,
    pub commitments: &'a [u8],
    pub proof: &'a [u8],
}

/// Verdict of `verify_txs` for one serialized transaction.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum TxVerdict {
    /// transaction ID, signature and cloak proof hold
    Accepted,
    /// the reference's `Err`
    Rejected,
    /// the transaction uses more of the VM than the payment subset: run it through `Tx::verify` on the CPU
    OutsideSubset,
}

/// A batch queued with `submit`; redeemed exactly once with `wait`.
#[derive(Debug)]
pub struct Ticket {
    id: u64,
    batch: usize,
}

/// A call of serialized transactions in flight (`GpuVerifier::submit_txs`).  It owns the bytes the library's engine thread
/// reads until the call has been waited for, and it BORROWS the verifier: a call that is dropped without `wait_txs` -- an
/// early `?`, a panic unwinding, a caller that discards it -- is waited for in `Drop` (verdicts thrown away), so the bytes
/// are never freed under a round that still reads them, and the library's record of the call does not pile up.
pub struct TxCall<'v> {
    verifier: &'v GpuVerifier,
    id: u64,
    n: usize,
    waited: bool,
    _blob: Vec<u8>,
    _offsets: Vec<u64>,
}

impl Drop for TxCall<'_> {
    fn drop(&mut self) {
        if !self.waited {
            let mut bitmap = vec![0u8; (self.n + 7) / 8];
            let mut status = vec![1u8; self.n];
            // (the return code does not matter here: after this call the library no longer reads `_blob` / `_offsets`)
            let _ = unsafe { sys::zkgpu_tx_verify_wait(self.verifier.v, self.id, bitmap.as_mut_ptr(), status.as_mut_ptr()) };
        }
    }
}

struct Generators {
    ps: *mut sys::zkgpu_pointset,
    capacity: usize,
}

/// Mirror of the reference's `Verifier` for whole blocks on one GPU.
pub struct GpuVerifier {
    ctx: *mut sys::zkgpu_ctx,
    gens: Generators,
    v: *mut sys::zkgpu_verifier,
    warning: Option<String>,
}

// the library serialises calls on a verifier with its own mutex, and hands out its error text as a per-thread copy made
// under that mutex (zkgpu_verifier_last_error); the handles are plain pointers to its heap objects
unsafe impl Send for GpuVerifier {}
unsafe impl Sync for GpuVerifier {}

impl GpuVerifier {
    /// `device`: HIP device index.  `gens_capacity`: `BulletproofGens::new(gens_capacity, 1)` -- 256 serves cloaks up to
    /// 2-in/2-out, 512 up to 4-in/4-out.  `lanes`: device batches in flight (0 = the library's default, 6).
    /// Derives `PedersenGens::default()` and the `BulletproofGens` chain on the device, builds the fixed-base tables
    /// (width chosen from the device's free memory) -- seconds, once per process.
    pub fn new(device: i32, gens_capacity: usize, lanes: i32) -> Result<Self, Error> {
        let abi = unsafe { sys::zkgpu_abi_version() };
        if abi != 3 {
            return Err(Error::AbiMismatch(abi));
        }
        // the library never edits the environment: it recommends, the host exports (a no-op once HIP has started or the
        // application has set the variable itself; call `apply_runtime_hint()` earlier if the application touches HIP first)
        apply_runtime_hint();
        let mut ctx: *mut sys::zkgpu_ctx = ptr::null_mut();
        check(unsafe { sys::zkgpu_init(device as c_int, &mut ctx) }, String::new)?;
        let ctx_err = |c: *mut sys::zkgpu_ctx| move || text(unsafe { sys::zkgpu_last_error(c) });
        let mut points = vec![0u8; 32 * (2 + 2 * gens_capacity)];
        let built = (|| {
            {
                let (b, rest) = points.split_at_mut(32);
                check(unsafe { sys::zkgpu_pedersen_gens(ctx, b.as_mut_ptr(), rest.as_mut_ptr()) }, ctx_err(ctx))?;
            }
            {
                let (g, h) = points[64..].split_at_mut(32 * gens_capacity);
                check(unsafe { sys::zkgpu_bulletproof_gens(ctx, gens_capacity, 0, g.as_mut_ptr(), h.as_mut_ptr()) }, ctx_err(ctx))?;
            }
            let mut ps: *mut sys::zkgpu_pointset = ptr::null_mut();
            check(unsafe { sys::zkgpu_pointset_create(ctx, points.as_ptr(), points.len() / 32, &mut ps) }, ctx_err(ctx))?;
            if let Err(e) = check(unsafe { sys::zkgpu_pointset_build_tables(ctx, ps, 0) }, ctx_err(ctx)) {
                unsafe { sys::zkgpu_pointset_destroy(ps) };
                return Err(e);
            }
            let mut v: *mut sys::zkgpu_verifier = ptr::null_mut();
            let rc = unsafe { sys::zkgpu_verifier_create(ctx, ps, gens_capacity, lanes as c_int, &mut v) };
            // ZKGPU_WSECOND_VERIFIER: created, with a warning (another verifier is alive on this device): `warning()` has the text
            if rc != sys::ZKGPU_WSECOND_VERIFIER {
                if let Err(e) = check(rc, ctx_err(ctx)) {
                    unsafe { sys::zkgpu_pointset_destroy(ps) };
                    return Err(e);
                }
            }
            Ok((ps, v, rc == sys::ZKGPU_WSECOND_VERIFIER))
        })();
        match built {
            Ok((ps, v, warned)) => {
                let warning = if warned { Some(text(unsafe { sys::zkgpu_verifier_last_error(v) })) } else { None };
                Ok(GpuVerifier { ctx, gens: Generators { ps, capacity: gens_capacity }, v, warning })
            }
            Err(e) => {
                unsafe { sys::zkgpu_destroy(ctx) };
                Err(e)
            }
        }
    }

    fn err(&self) -> impl FnOnce() -> String + '_ {
        move || text(unsafe { sys::zkgpu_verifier_last_error(self.v) })
    }

    /// Set when the verifier was created beside another live one on the same device (`ZKGPU_WSECOND_VERIFIER`): it works,
    /// but one verifier per process and device is the shape the library's queue budget is made for.
    pub fn warning(&self) -> Option<&str> {
        self.warning.as_deref()
    }

    pub fn gens_capacity(&self) -> usize {
        self.gens.capacity
    }

    /// Window width and bytes of the generator tables resident on the device.
    pub fn table_info(&self) -> (i32, usize) {
        unsafe { (sys::zkgpu_pointset_table_bits(self.gens.ps) as i32, sys::zkgpu_pointset_table_bytes(self.gens.ps)) }
    }

    /// (lanes in use, lanes asked for, lanes dropped at creation, HIP runtime started before GPU_MAX_HW_QUEUES was set)
    pub fn queue_info(&self) -> Result<(i32, i32, i32, bool), Error> {
        let mut out = [0 as c_int; 4];
        check(unsafe { sys::zkgpu_verifier_queue_info(self.v, out.as_mut_ptr()) }, self.err())?;
        Ok((out[0] as i32, out[1] as i32, out[2] as i32, out[3] != 0))
    }

    /// Drop-in for `stmts.iter().map(|s| r1cs::Verifier::verify(..))` over a block of any mix of shapes: one verdict per
    /// statement, in order.  `randomness`: 64 bytes per statement for the random weights of the batched identity test, or
    /// `None` for the operating system's.  `Err` = no verdicts at all (treat every transaction as unverified).
    pub fn verify_block(&self, stmts: &[CloakStatement<'_>], randomness: Option<&[u8]>) -> Result<Vec<bool>, Error> {
        let n = stmts.len();
        if n == 0 {
            return Ok(Vec::new());
        }
        let mut n_in = Vec::with_capacity(n);
        let mut n_out = Vec::with_capacity(n);
        let mut com = Vec::new();
        let mut proofs = Vec::new();
        let mut offs = Vec::with_capacity(n + 1);
        offs.push(0u64);
        for s in stmts {
            if s.commitments.len() != 64 * (s.n_in as usize + s.n_out as usize) {
                return Err(Error::InvalidArgument("commitments: 64 bytes per value".into()));
            }
            n_in.push(s.n_in);
            n_out.push(s.n_out);
            com.extend_from_slice(s.commitments);
            proofs.extend_from_slice(s.proof);
            offs.push(proofs.len() as u64);
        }
        if let Some(r) = randomness {
            if r.len() != 64 * n {
                return Err(Error::InvalidArgument("randomness: 64 bytes per statement".into()));
            }
        }
        let mut bitmap = vec![0u8; (n + 7) / 8];
        check(
            unsafe {
                sys::zkgpu_verifier_verify(
                    self.v,
                    n,
                    n_in.as_ptr(),
                    n_out.as_ptr(),
                    com.as_ptr(),
                    proofs.as_ptr(),
                    offs.as_ptr(),
                    randomness.map_or(ptr::null(), |r| r.as_ptr()),
                    bitmap.as_mut_ptr(),
                )
            },
            self.err(),
        )?;
        Ok(bits(&bitmap, n))
    }

    /// The serialized-transaction format `verify_txs` reads is a RECOLLECTION of the ZkVM notes: until
    /// a maintainer has compared it with the real `zkvm` crate and calls this, every transaction is `OutsideSubset`.
    pub fn enable_recollected_tx_format(&self) -> Result<(), Error> {
        check(unsafe { sys::zkgpu_verifier_set_tx_format(self.v, sys::ZKGPU_TXFORMAT_RECOLLECTED_V1) }, self.err())
    }

    /// Drop-in for `txs.iter().map(|tx| tx.verify(bp_gens))` on `Tx::encode()` bytes (payment subset): VM, transaction
    /// ID, MuSig / Schnorr signature and cloak proof.  `host_threads`: 0 = the CPUs the process may keep busy.
    pub fn verify_txs(&self, txs: &[&[u8]], host_threads: i32) -> Result<Vec<TxVerdict>, Error> {
        let n = txs.len();
        if n == 0 {
            return Ok(Vec::new());
        }
        let mut blob = Vec::with_capacity(txs.iter().map(|t| t.len()).sum());
        let mut offs = Vec::with_capacity(n + 1);
        offs.push(0u64);
        for t in txs {
            blob.extend_from_slice(t);
            offs.push(blob.len() as u64);
        }
        let mut bitmap = vec![0u8; (n + 7) / 8];
        let mut status = vec![1u8; n];
        check(
            unsafe {
                sys::zkgpu_tx_verify_batch(self.v, n, blob.as_ptr(), offs.as_ptr(), host_threads as c_int, bitmap.as_mut_ptr(), status.as_mut_ptr())
            },
            self.err(),
        )?;
        Ok((0..n)
            .map(|i| match (status[i], (bitmap[i / 8] >> (i % 8)) & 1) {
                (0, 1) => TxVerdict::Accepted,
                (2, _) => TxVerdict::OutsideSubset,
                _ => TxVerdict::Rejected, // (status 0 without its accept bit cannot happen; it would be a rejection)
            })
            .collect())
    }

    /// `verify_txs` in two halves, callable from many threads at once (upstream's `Tx::verify` is pure and `&self`): the
    /// call is queued and this returns; an engine thread of the verifier merges whatever is queued into rounds (dynamic
    /// batching), so that eight threads handing over 1024 transactions each see the rate of calls of several thousand.
    pub fn submit_txs(&self, txs: &[&[u8]], host_threads: i32) -> Result<TxCall<'_>, Error> {
        let n = txs.len();
        if n == 0 {
            return Err(Error::InvalidArgument("submit_txs: no transactions".into()));
        }
        let mut blob = Vec::with_capacity(txs.iter().map(|t| t.len()).sum());
        let mut offsets = Vec::with_capacity(n + 1);
        offsets.push(0u64);
        for t in txs {
            blob.extend_from_slice(t);
            offsets.push(blob.len() as u64);
        }
        let mut id = 0u64;
        // (the heap buffers of `blob` and `offsets` do not move when the vectors are moved into the handle)
        check(unsafe { sys::zkgpu_tx_verify_submit(self.v, n, blob.as_ptr(), offsets.as_ptr(), host_threads as c_int, &mut id) }, self.err())?;
        Ok(TxCall { verifier: self, id, n, waited: false, _blob: blob, _offsets: offsets })
    }

    /// Blocks until the call's round is done: one verdict per transaction of THAT call.
    pub fn wait_txs(&self, mut call: TxCall<'_>) -> Result<Vec<TxVerdict>, Error> {
        if !ptr::eq(call.verifier, self) {
            return Err(Error::InvalidArgument("wait_txs: the call was submitted to another verifier".into())); // (its Drop waits there)
        }
        let n = call.n;
        let mut bitmap = vec![0u8; (n + 7) / 8];
        let mut status = vec![1u8; n];
        let rc = unsafe { sys::zkgpu_tx_verify_wait(self.v, call.id, bitmap.as_mut_ptr(), status.as_mut_ptr()) };
        call.waited = true; // (whatever it returned, the library is done with the call's bytes)
        check(rc, self.err())?;
        Ok((0..n)
            .map(|i| match (status[i], (bitmap[i / 8] >> (i % 8)) & 1) {
                (0, 1) => TxVerdict::Accepted,
                (2, _) => TxVerdict::OutsideSubset,
                _ => TxVerdict::Rejected,
            })
            .collect())
    }

    /// Transactions per merged device batch (default 4096; the MI355X bench uses 10 240).
    pub fn set_merge(&self, transactions: usize) -> Result<(), Error> {
        check(unsafe { sys::zkgpu_verifier_set_merge(self.v, transactions) }, self.err())
    }

    /// Sizes every lane's workspace for device batches of `transactions` statements of the shape beforehand (otherwise
    /// workspaces grow on demand, and `hipMalloc` synchronises the device): for verifiers whose batches vary in shape and size.
    pub fn reserve(&self, n_in: u32, n_out: u32, transactions: usize) -> Result<(), Error> {
        check(unsafe { sys::zkgpu_verifier_reserve(self.v, n_in, n_out, transactions) }, self.err())
    }

    /// Queues `batch` statements of ONE shape from host memory and returns at once; the slices are free again on return
    /// (they are copied into pinned staging memory).  Batches in flight are merged into device batches.
    pub fn submit(&self, n_in: u32, n_out: u32, batch: usize, commitments: &[u8], proofs: &[u8], proof_len: usize, randomness: Option<&[u8]>) -> Result<Ticket, Error> {
        if batch == 0 || commitments.len() != batch * 64 * (n_in as usize + n_out as usize) || proofs.len() != batch * proof_len {
            return Err(Error::InvalidArgument("submit: lengths do not match the batch".into()));
        }
        if let Some(r) = randomness {
            if r.len() != 64 * batch {
                return Err(Error::InvalidArgument("randomness: 64 bytes per statement".into()));
            }
        }
        let mut id = 0u64;
        check(
            unsafe {
                sys::zkgpu_verifier_submit(
                    self.v,
                    n_in,
                    n_out,
                    batch,
                    commitments.as_ptr(),
                    proofs.as_ptr(),
                    proof_len,
                    randomness.map_or(ptr::null(), |r| r.as_ptr()),
                    &mut id,
                )
            },
            self.err(),
        )?;
        Ok(Ticket { id, batch })
    }

    /// Blocks until the ticket's device batch is done: one verdict per statement of THAT ticket.
    pub fn wait(&self, ticket: Ticket) -> Result<Vec<bool>, Error> {
        let mut bitmap = vec![0u8; (ticket.batch + 7) / 8];
        check(unsafe { sys::zkgpu_verifier_wait(self.v, ticket.id, bitmap.as_mut_ptr()) }, self.err())?;
        Ok(bits(&bitmap, ticket.batch))
    }

    /// One process per GPU: every rank holds the whole block, verifies its shard (contiguous, balanced by
    /// multiscalar-multiplication terms) and receives every shard's verdicts through one `ncclAllGather`.
    pub fn verify_block_sharded(&self, comm: &Comm, stmts: &[CloakStatement<'_>], randomness: Option<&[u8]>) -> Result<Vec<bool>, Error> {
        let n = stmts.len();
        if n == 0 {
            return Ok(Vec::new());
        }
        let n_in: Vec<u32> = stmts.iter().map(|s| s.n_in).collect();
        let n_out: Vec<u32> = stmts.iter().map(|s| s.n_out).collect();
        let mut com = Vec::new();
        let mut proofs = Vec::new();
        let mut offs = vec![0u64];
        // the library reads 64 * (n_in + n_out) bytes per statement and 64 bytes of randomness per statement: the same
        // checks as `verify_block`, or a short slice would be an out-of-bounds read behind a safe fn
        if let Some(r) = randomness {
            if r.len() != 64 * n {
                return Err(Error::InvalidArgument("randomness: 64 bytes per statement".into()));
            }
        }
        for s in stmts {
            if s.commitments.len() != 64 * (s.n_in as usize + s.n_out as usize) {
                return Err(Error::InvalidArgument("commitments: 64 bytes per value".into()));
            }
            com.extend_from_slice(s.commitments);
            proofs.extend_from_slice(s.proof);
            offs.push(proofs.len() as u64);
        }
        let mut bitmap = vec![0u8; (n + 7) / 8];
        check(
            unsafe {
                sys::zkgpu_verifier_verify_sharded(
                    self.v,
                    comm.raw,
                    n,
                    n_in.as_ptr(),
                    n_out.as_ptr(),
                    com.as_ptr(),
                    proofs.as_ptr(),
                    offs.as_ptr(),
                    randomness.map_or(ptr::null(), |r| r.as_ptr()),
                    bitmap.as_mut_ptr(),
                )
            },
            self.err(),
        )?;
        Ok(bits(&bitmap, n))
    }

    /// The communicator of this process's GPU.  `id`: 128 bytes from `Comm::unique_id()` of rank 0, shipped to the other
    /// ranks over whatever channel the node already has.
    pub fn comm(&self, rank: i32, world: i32, id: &[u8; sys::ZKGPU_COMM_ID_BYTES]) -> Result<Comm, Error> {
        let mut raw: *mut sys::zkgpu_comm = ptr::null_mut();
        let ctx = self.ctx;
        check(unsafe { sys::zkgpu_comm_create(ctx, rank as c_int, world as c_int, id.as_ptr(), &mut raw) }, move || text(unsafe { sys::zkgpu_last_error(ctx) }))?;
        Ok(Comm { raw })
    }

    /// One multiscalar multiplication (`RistrettoPoint::vartime_multiscalar_mul` on compressed points): 32-byte
    /// little-endian scalars, 32-byte encodings -> the encoding of the sum.
    pub fn msm(&self, scalars: &[u8], points: &[u8]) -> Result<[u8; 32], Error> {
        if scalars.len() != points.len() || scalars.len() % 32 != 0 {
            return Err(Error::InvalidArgument("msm: 32 bytes per scalar and per point".into()));
        }
        let mut out = [0u8; 32];
        let mut bad = 0usize;
        let ctx = self.ctx;
        check(unsafe { sys::zkgpu_msm(ctx, scalars.as_ptr(), points.as_ptr(), scalars.len() / 32, out.as_mut_ptr(), &mut bad) }, move || {
            text(unsafe { sys::zkgpu_last_error(ctx) })
        })?;
        Ok(out)
    }
}

impl Drop for GpuVerifier {
    fn drop(&mut self) {
        unsafe {
            sys::zkgpu_verifier_destroy(self.v); // finishes what is in flight
            sys::zkgpu_pointset_destroy(self.gens.ps);
            sys::zkgpu_destroy(self.ctx);
        }
    }
}

/// RCCL communicator of one rank (one process per GPU); the only collective of the sharded verification is the
/// all-gather of the accept bitmaps.
pub struct Comm {
    raw: *mut sys::zkgpu_comm,
}
unsafe impl Send for Comm {}

impl Comm {
    /// Rank 0 makes the id and ships it to the others.
    pub fn unique_id() -> Result<[u8; sys::ZKGPU_COMM_ID_BYTES], Error> {
        let mut id = [0u8; sys::ZKGPU_COMM_ID_BYTES];
        check(unsafe { sys::zkgpu_comm_unique_id(id.as_mut_ptr()) }, || "RCCL could not be loaded".to_string())?;
        Ok(id)
    }
}

impl Drop for Comm {
    fn drop(&mut self) {
        unsafe { sys::zkgpu_comm_destroy(self.raw) };
    }
}

/// Text of a status code (`zkgpu_strerror`).
/// `zkgpu_runtime_hint`: exports what the library recommends ("GPU_MAX_HW_QUEUES=18") with `std::env::set_var` when the
/// answer is `ZKGPU_HINT_APPLY` (variable unset, HIP runtime not started).  Returns the library's answer
/// (0 applied, 1 already present, 2 too late: the process runs on the runtime's default, reported by `queue_info`).
/// Call it before the process's first HIP call, on the main thread before other threads read the environment.
pub fn apply_runtime_hint() -> i32 {
    let mut buf = [0 as std::os::raw::c_char; 128];
    let rc = unsafe { sys::zkgpu_runtime_hint(buf.as_mut_ptr(), buf.len()) };
    if rc == sys::ZKGPU_HINT_APPLY as i32 {
        let text = unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
        for pair in text.split('\n') {
            if let Some((name, value)) = pair.split_once('=') {
                std::env::set_var(name, value);
            }
        }
    }
    rc
}

pub fn strerror(code: i32) -> String {
    text(unsafe { sys::zkgpu_strerror(code as c_int) })
}
