"""ctypes binding of oracle/_build/liboracle.so (TEST INFRASTRUCTURE ONLY).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never from zkvm_amd/.  `load()` builds the library with oracle/Makefile when the
.so is missing or older than its sources.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "liboracle.so")
_lib = None


class Fe(C.Structure):
    _fields_ = [("v", C.c_uint64 * 5)]


class Sc(C.Structure):
    _fields_ = [("v", C.c_uint64 * 4)]


class Ge(C.Structure):
    _fields_ = [("X", Fe), ("Y", Fe), ("Z", Fe), ("T", Fe)]


class Transcript(C.Structure):
    _fields_ = [("st", C.c_uint8 * 200), ("pos", C.c_uint8), ("pos_begin", C.c_uint8), ("cur_flags", C.c_uint8)]


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".h"))]
    stale = force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", HERE], check=True, capture_output=True)
    return SO


def load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(SO)
        _lib.zko_msm.restype = C.c_int
        _lib.zko_verify_batch.restype = C.c_int
        _lib.zko_max_threads.restype = C.c_int
    return _lib


# ---- field ------------------------------------------------------------------
def fe_from_int(x: int) -> Fe:
    f = Fe()
    load().fe_frombytes(C.byref(f), (x % (2**255)).to_bytes(32, "little"))
    return f


def fe_to_int(f: Fe) -> int:
    out = C.create_string_buffer(32)
    load().fe_tobytes(out, C.byref(f))
    return int.from_bytes(out.raw, "little")


def fe_binop(name: str, a: int, b: int) -> int:
    fa, fb, r = fe_from_int(a), fe_from_int(b), Fe()
    getattr(load(), name)(C.byref(r), C.byref(fa), C.byref(fb))
    return fe_to_int(r)


def fe_unop(name: str, a: int) -> int:
    fa, r = fe_from_int(a), Fe()
    getattr(load(), name)(C.byref(r), C.byref(fa))
    return fe_to_int(r)


def fe_sqrt_ratio_m1(u: int, v: int) -> Tuple[bool, int]:
    fu, fv, r = fe_from_int(u), fe_from_int(v), Fe()
    ok = load().fe_sqrt_ratio_m1(C.byref(r), C.byref(fu), C.byref(fv))
    return bool(ok), fe_to_int(r)


# ---- scalars ----------------------------------------------------------------
def sc_from_int(x: int) -> Sc:
    s = Sc()
    load().sc_from_bytes_wide(C.byref(s), (x % (2**512)).to_bytes(64, "little"))
    return s


def sc_to_int(s: Sc) -> int:
    out = C.create_string_buffer(32)
    load().sc_to_bytes(out, C.byref(s))
    return int.from_bytes(out.raw, "little")


def sc_binop(name: str, a: int, b: int) -> int:
    sa, sb, r = sc_from_int(a), sc_from_int(b), Sc()
    getattr(load(), name)(C.byref(r), C.byref(sa), C.byref(sb))
    return sc_to_int(r)


def sc_invert(a: int) -> int:
    sa, r = sc_from_int(a), Sc()
    load().sc_invert(C.byref(r), C.byref(sa))
    return sc_to_int(r)


def sc_reduce_wide(b: bytes) -> int:
    s = Sc()
    load().sc_from_bytes_wide(C.byref(s), b)
    return sc_to_int(s)


# ---- group ------------------------------------------------------------------
def decode(b: bytes) -> Optional[Ge]:
    p = Ge()
    return p if load().ristretto_decode(C.byref(p), b) else None


def encode(p: Ge) -> bytes:
    out = C.create_string_buffer(32)
    load().ristretto_encode(out, C.byref(p))
    return out.raw


def from_uniform_bytes(b: bytes) -> bytes:
    p = Ge()
    load().ristretto_from_uniform_bytes(C.byref(p), b)
    return encode(p)


def basepoint() -> Ge:
    p = Ge()
    load().ge_basepoint(C.byref(p))
    return p


def scalarmult(k: int, p: Ge) -> Ge:
    r, s = Ge(), sc_from_int(k)
    load().ge_scalarmult(C.byref(r), C.byref(s), C.byref(p))
    return r


def add(p: Ge, q: Ge) -> Ge:
    r = Ge()
    load().ge_add(C.byref(r), C.byref(p), C.byref(q))
    return r


def double(p: Ge) -> Ge:
    r = Ge()
    load().ge_double(C.byref(r), C.byref(p))
    return r


def msm_points(kind: str, scalars: Sequence[int], points: Sequence[Ge]) -> Ge:
    n = len(scalars)
    sa = (Sc * max(n, 1))(*[sc_from_int(k) for k in scalars])
    pa = (Ge * max(n, 1))(*points)
    r = Ge()
    getattr(load(), "ge_msm_" + kind)(C.byref(r), sa, pa, C.c_size_t(n))
    return r


# ---- byte-level entry points (the shapes of include/zkgpu.h) ------------------
def msm(scalars: bytes, points: bytes) -> Tuple[int, bytes, int]:
    """-> (rc, 32-byte encoding, bad_index)"""
    n = len(scalars) // 32
    assert len(scalars) == 32 * n and len(points) == 32 * n
    out = C.create_string_buffer(32)
    bad = C.c_size_t(0)
    rc = load().zko_msm(scalars, points, C.c_size_t(n), out, C.byref(bad))
    return rc, out.raw, bad.value


def verify_batch(scalars: bytes, points: bytes, offsets: Sequence[int], threads: int = 1) -> bytes:
    b = len(offsets) - 1
    off = (C.c_uint64 * (b + 1))(*offsets)
    bitmap = C.create_string_buffer((b + 7) // 8 or 1)
    rc = load().zko_verify_batch(scalars, points, off, C.c_size_t(b), bitmap, C.c_int(threads))
    assert rc == 0
    return bitmap.raw[: (b + 7) // 8]


def decode_batch(points: bytes) -> bytes:
    n = len(points) // 32
    ok = C.create_string_buffer(max(n, 1))
    load().zko_decode_batch(points, C.c_size_t(n), ok)
    return ok.raw[:n]


def max_threads() -> int:
    return load().zko_max_threads()


# ---- hashing / transcripts ---------------------------------------------------
def sha3_512(data: bytes) -> bytes:
    out = C.create_string_buffer(64)
    load().sha3_512(out, data, C.c_size_t(len(data)))
    return out.raw


class Shake256Ctx(C.Structure):
    _fields_ = [("st", C.c_uint64 * 25), ("pos", C.c_uint), ("squeezing", C.c_int)]


def shake256(data: bytes, outlen: int) -> bytes:
    c = Shake256Ctx()
    lib = load()
    lib.shake256_init(C.byref(c))
    lib.shake256_absorb(C.byref(c), data, C.c_size_t(len(data)))
    out = C.create_string_buffer(outlen)
    # squeeze in two pieces to exercise the streaming path
    h = outlen // 2
    lib.shake256_squeeze(C.byref(c), out, C.c_size_t(h))
    lib.shake256_squeeze(C.byref(c), C.byref(out, h), C.c_size_t(outlen - h))
    return out.raw


class MerlinTranscript:
    def __init__(self, label: bytes):
        self.t = Transcript()
        load().merlin_init(C.byref(self.t), label, C.c_size_t(len(label)))

    def append_message(self, label: bytes, msg: bytes) -> None:
        load().merlin_append_message(C.byref(self.t), label, msg, C.c_size_t(len(msg)))

    def append_u64(self, label: bytes, x: int) -> None:
        load().merlin_append_u64(C.byref(self.t), label, C.c_uint64(x))

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        out = C.create_string_buffer(n)
        load().merlin_challenge_bytes(C.byref(self.t), label, out, C.c_size_t(n))
        return out.raw

    def challenge_scalar(self, label: bytes) -> int:
        s = Sc()
        load().merlin_challenge_scalar(C.byref(self.t), label, C.byref(s))
        return sc_to_int(s)


def pedersen_gens() -> Tuple[bytes, bytes]:
    b, bb = Ge(), Ge()
    load().pedersen_gens(C.byref(b), C.byref(bb))
    return encode(b), encode(bb)


def bulletproof_gens(n: int, which: str, party: int = 0) -> List[bytes]:
    arr = (Ge * n)()
    load().bulletproof_gens_chain(arr, C.c_size_t(n), C.c_char(which.encode()), C.c_uint32(party))
    return [encode(arr[i]) for i in range(n)]


# ---- R1CS / cloak (oracle/r1cs.c, oracle/cloak.c) ------------------------------------
class R1csMsm(C.Structure):
    _fields_ = [("dyn_scalars", C.POINTER(C.c_uint8)), ("dyn_points", C.POINTER(C.c_uint8)),
                ("static_scalars", C.POINTER(C.c_uint8)), ("n_dyn", C.c_size_t), ("n_static", C.c_size_t),
                ("padded_n", C.c_size_t)]


def cloak_proof_size(padded_n: int) -> int:
    lib = load()
    lib.r1cs_proof_size.restype = C.c_size_t
    return int(lib.r1cs_proof_size(C.c_size_t(padded_n)))


def cloak_prove(q: Sequence[int], flavors: Sequence[bytes], n_in: int, n_out: int, seed: bytes):
    """-> (rc, commitments[64 * (n_in + n_out)], proof bytes, multipliers)"""
    nv = n_in + n_out
    assert len(q) == nv and len(flavors) == nv and len(seed) == 32
    qa = (C.c_uint64 * nv)(*q)
    com = C.create_string_buffer(64 * nv)
    cap = 1 + 32 * (14 + 2 * 16 + 2)
    proof = C.create_string_buffer(cap)
    plen, nm = C.c_size_t(0), C.c_size_t(0)
    rc = load().zko_cloak_prove(qa, b"".join(flavors), C.c_size_t(n_in), C.c_size_t(n_out), seed, com, proof,
                                C.c_size_t(cap), C.byref(plen), C.byref(nm))
    return rc, com.raw, proof.raw[: plen.value], nm.value


def cloak_prove_batch(count: int, n_in: int, n_out: int, seed: bytes, threads: int = 1):
    """-> (commitments: count x 64*(n_in+n_out) bytes, proofs: list of bytes)"""
    nv = n_in + n_out
    stride = 1 + 32 * (14 + 2 * 16 + 2)
    com = C.create_string_buffer(64 * nv * count)
    proofs = C.create_string_buffer(stride * count)
    plen = C.c_size_t(0)
    rc = load().zko_cloak_prove_batch(C.c_size_t(count), C.c_size_t(n_in), C.c_size_t(n_out), seed, com, proofs,
                                      C.c_size_t(stride), C.byref(plen), C.c_int(threads))
    assert rc == 0, "oracle prover failed"
    n = plen.value
    return com.raw, [proofs.raw[i * stride: i * stride + n] for i in range(count)]


def cloak_verify(commitments: bytes, n_in: int, n_out: int, proof: bytes, r_bytes: bytes) -> bool:
    assert len(r_bytes) == 64
    return bool(load().zko_cloak_verify(commitments, C.c_size_t(n_in), C.c_size_t(n_out), proof,
                                        C.c_size_t(len(proof)), r_bytes))


def cloak_verify_prepare(commitments: bytes, n_in: int, n_out: int, proof: bytes, r_bytes: bytes):
    """-> None when the proof is malformed, else (dyn_scalars, dyn_points, static_scalars, padded_n)"""
    m = R1csMsm()
    lib = load()
    rc = lib.zko_cloak_verify_prepare(commitments, C.c_size_t(n_in), C.c_size_t(n_out), proof, C.c_size_t(len(proof)),
                                      r_bytes, C.byref(m))
    if rc != 0:
        return None
    out = (C.string_at(m.dyn_scalars, 32 * m.n_dyn), C.string_at(m.dyn_points, 32 * m.n_dyn),
           C.string_at(m.static_scalars, 32 * m.n_static), int(m.padded_n))
    lib.r1cs_msm_free(C.byref(m))
    return out


def cloak_verify_challenges(commitments: bytes, n_in: int, n_out: int, proof: bytes, r_bytes: bytes):
    """-> None when the proof is malformed, else the verifier's challenges in transcript order as ints:
    second-phase challenges, y, z, u, x, w, the k inner-product challenges"""
    cap = 128
    buf = C.create_string_buffer(32 * cap)
    n = C.c_size_t(0)
    rc = load().zko_cloak_verify_challenges(commitments, C.c_size_t(n_in), C.c_size_t(n_out), proof, C.c_size_t(len(proof)),
                                            r_bytes, buf, C.c_size_t(cap), C.byref(n))
    if rc != 0:
        return None
    return [int.from_bytes(buf.raw[32 * i: 32 * i + 32], "little") for i in range(n.value)]


def cloak_verify_batch(commitments: bytes, n_in: int, n_out: int, proofs: bytes, proof_len: int, r_bytes: bytes,
                       threads: int = 1) -> bytes:
    """Full CPU verification (transcript replay + MSM) of count = len(proofs) / proof_len proofs; -> accept bytes"""
    count = len(proofs) // proof_len
    acc = C.create_string_buffer(max(count, 1))
    load().zko_cloak_verify_batch(C.c_size_t(count), C.c_size_t(n_in), C.c_size_t(n_out), commitments, proofs,
                                  C.c_size_t(proof_len), C.c_size_t(proof_len), r_bytes, acc, C.c_int(threads))
    return acc.raw[:count]


L_ORDER = 2**252 + 27742317777372353535851937790883648493

# ---- statements other than the cloak (oracle/gadgets.c) ----------------------------------------
GADGET_RANGE, GADGET_SHUFFLE = 1, 2


def gadget_commitments(kind: int, param: int) -> int:
    lib = load()
    lib.zko_gadget_commitments.restype = C.c_size_t
    return int(lib.zko_gadget_commitments(C.c_int(kind), C.c_size_t(param)))


def gadget_prove(kind: int, param: int, values: Sequence[int], seed: bytes):
    """-> (rc, commitments m x 32 bytes, proof)"""
    m = gadget_commitments(kind, param)
    assert len(values) == m and len(seed) == 32
    vals = b"".join(int(v % L_ORDER).to_bytes(32, "little") for v in values)
    com = C.create_string_buffer(32 * m)
    cap = 1 + 32 * (16 + 2 * 20)
    proof = C.create_string_buffer(cap)
    plen = C.c_size_t(0)
    rc = load().zko_gadget_prove(C.c_int(kind), C.c_size_t(param), vals, seed, com, proof, C.c_size_t(cap), C.byref(plen))
    return rc, com.raw, proof.raw[: plen.value]


def gadget_verify(kind: int, param: int, commitments: bytes, proof: bytes, r_bytes: bytes) -> bool:
    return bool(load().zko_gadget_verify(C.c_int(kind), C.c_size_t(param), commitments, proof, C.c_size_t(len(proof)), r_bytes))


def gadget_verify_prepare(kind: int, param: int, commitments: bytes, proof: bytes, r_bytes: bytes):
    """-> None when malformed, else (dyn_scalars, dyn_points, static_scalars, padded_n, challenges as ints)"""
    m = R1csMsm()
    lib = load()
    cap = 128
    buf = C.create_string_buffer(32 * cap)
    n = C.c_size_t(0)
    rc = lib.zko_gadget_verify_prepare(C.c_int(kind), C.c_size_t(param), commitments, proof, C.c_size_t(len(proof)), r_bytes,
                                       C.byref(m), buf, C.c_size_t(cap), C.byref(n))
    if rc != 0:
        return None
    out = (C.string_at(m.dyn_scalars, 32 * m.n_dyn), C.string_at(m.dyn_points, 32 * m.n_dyn),
           C.string_at(m.static_scalars, 32 * m.n_static), int(m.padded_n),
           [int.from_bytes(buf.raw[32 * i: 32 * i + 32], "little") for i in range(n.value)])
    lib.r1cs_msm_free(C.byref(m))
    return out


# ---- ZkVM transactions of the payment subset (zkvm_tx.c) --------------------------------
def tx_build_payment(n_in: int, n_out: int, quantities: Sequence[int], flavors: Sequence[bytes], seed: bytes,
                     mintime: int = 0, maxtime: int = 2 ** 63) -> bytes:
    """a signed transaction: n_in unspent contracts -> cloak -> n_out contracts (b"" when it cannot be built)"""
    lib = load()
    lib.zko_tx_build_payment.restype = C.c_size_t
    nv = n_in + n_out
    assert len(quantities) == nv and len(flavors) == nv and len(seed) == 32
    q = (C.c_uint64 * nv)(*quantities)
    cap = 65536
    out = C.create_string_buffer(cap)
    n = lib.zko_tx_build_payment(C.c_size_t(n_in), C.c_size_t(n_out), q, b"".join(flavors), seed, C.c_uint64(mintime),
                                 C.c_uint64(maxtime), out, C.c_size_t(cap))
    return out.raw[:n]


def tx_wrap_payment(n_in: int, n_out: int, commitments: bytes, proof: bytes, seed: bytes, mintime: int = 0,
                    maxtime: int = 2 ** 63) -> bytes:
    """a signed transaction around an existing cloak proof (no proving)"""
    lib = load()
    lib.zko_tx_wrap_payment.restype = C.c_size_t
    assert len(commitments) == 64 * (n_in + n_out) and len(seed) == 32
    cap = 65536
    out = C.create_string_buffer(cap)
    n = lib.zko_tx_wrap_payment(C.c_size_t(n_in), C.c_size_t(n_out), commitments, proof, C.c_size_t(len(proof)), seed,
                                C.c_uint64(mintime), C.c_uint64(maxtime), out, C.c_size_t(cap))
    return out.raw[:n]


def tx_id(tx: bytes):
    """-> (status, txid, n_in, n_out); status 0 ok, 1 invalid, 2 outside the subset"""
    txid = C.create_string_buffer(32)
    a, b = C.c_size_t(0), C.c_size_t(0)
    rc = load().zko_tx_id(tx, C.c_size_t(len(tx)), txid, C.byref(a), C.byref(b))
    return rc, txid.raw, a.value, b.value


def tx_verify(tx: bytes, r_bytes: bytes) -> int:
    """Tx::verify: 0 accepted, 1 rejected, 2 outside the subset"""
    assert len(r_bytes) == 64
    return int(load().zko_tx_verify(tx, C.c_size_t(len(tx)), r_bytes))
