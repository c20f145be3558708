"""ZkVM transactions of the payment subset: the product's host half of
Tx::verify (zkvm_tx.hpp: wire format, VM, transaction ID, signature equation) against the oracle's independent
restatement (oracle/zkvm_tx.c), on transactions the oracle builds and signs."""
import ctypes as C
import hashlib
import os
import random
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host():
    from zkvm_amd.build import build, HOST_OUT
    build()
    return C.CDLL(HOST_OUT)


def payment(oracle, n_in, n_out, seed, two_flavors=False, mintime=0, maxtime=2 ** 40):
    rng = random.Random(seed)
    fl = [rng.randrange(2 ** 250).to_bytes(32, "little") for _ in range(2)]
    q_in = [rng.randrange(1, 2 ** 40) for _ in range(n_in)]
    f_in = [fl[j & 1] if two_flavors else fl[0] for j in range(n_in)]
    tot = [0, 0]
    for a, f in zip(q_in, f_in):
        tot[fl.index(f)] += a
    q_out, f_out = [], []
    for j in range(n_out):
        fi = (j & 1) if two_flavors else 0
        last = all(((jj & 1) if two_flavors else 0) != fi for jj in range(j + 1, n_out))
        a = tot[fi] if last else tot[fi] // 3
        tot[fi] -= a
        q_out.append(a); f_out.append(fl[fi])
    assert tot == [0, 0]
    tx = oracle.tx_build_payment(n_in, n_out, q_in + q_out, f_in + f_out, hashlib.sha256(b"tx %d" % seed).digest(), mintime, maxtime)
    assert tx
    return tx


def prepare(host, tx):
    txid = C.create_string_buffer(32)
    n_in, n_out = C.c_uint32(0), C.c_uint32(0)
    com = C.create_string_buffer(64 * 128)
    ss, sp = C.create_string_buffer(32 * 80), C.create_string_buffer(32 * 80)
    n_sig, po, pl = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    rc = host.zkhost_tx_prepare(tx, C.c_size_t(len(tx)), txid, C.byref(n_in), C.byref(n_out), com, C.c_size_t(64 * 128), ss, sp,
                                C.c_size_t(80), C.byref(n_sig), C.byref(po), C.byref(pl))
    return rc, txid.raw, n_in.value, n_out.value, com.raw[: 64 * (n_in.value + n_out.value)], ss.raw[: 32 * n_sig.value], \
        sp.raw[: 32 * n_sig.value], po.value, pl.value


@pytest.mark.parametrize("shape", [(1, 1), (2, 2), (3, 2), (1, 2), (6, 3)])     # (6, 3): more values and keys than a statement keeps inline
def test_transaction_statement_equals_oracle(host, oracle, shape):
    """Same transaction ID, same cloak statement, and a signature equation that the oracle's multiscalar multiplication
    finds to be the identity -- for a valid transaction; the same verdict for damaged ones."""
    n_in, n_out = shape
    tx = payment(oracle, n_in, n_out, 100 * n_in + n_out, two_flavors=n_in >= 2 and n_out >= 2)
    rc, txid, a, b, com, ss, sp, po, pl = prepare(host, tx)
    want_rc, want_id, wa, wb = oracle.tx_id(tx)
    assert (rc, txid, a, b) == (0, want_id, n_in, n_out) == (want_rc, want_id, wa, wb)
    proof = tx[po: po + pl]
    assert po + pl == len(tx)
    r = hashlib.shake_256(b"tx r").digest(64)
    assert oracle.cloak_verify(com, n_in, n_out, proof, r)          # the statement the VM extracted is the one that was proved
    n_terms = len(ss) // 32
    assert n_terms == 2 + n_in
    assert oracle.verify_batch(ss, sp, [0, n_terms]) == bytes([1])   # s B - R - sum c a_i X_i == identity
    assert oracle.tx_verify(tx, r) == 0
    # damage, one byte at a time: header, program (a contract), signature R, signature s, proof
    sig_at = po - 4 - 64
    for at in (8, 16, 40, 28 + 40, sig_at + 3, sig_at + 35, po + 50, len(tx) - 1):
        bad = bytearray(tx); bad[at] ^= 1
        bad = bytes(bad)
        want = oracle.tx_verify(bad, r)
        assert want == 1, at
        rc2, txid2, a2, b2, com2, ss2, sp2, po2, pl2 = prepare(host, bad)
        if rc2 != 0:
            assert rc2 == 1
            continue
        sig_ok = oracle.verify_batch(ss2, sp2, [0, len(ss2) // 32]) == bytes([1])
        proof_ok = oracle.cloak_verify(com2, a2, b2, bad[po2: po2 + pl2], r)
        assert not (sig_ok and proof_ok), at


def test_transactions_outside_the_subset_are_reported_not_rejected(host, oracle):
    tx = payment(oracle, 2, 2, 7)
    plen = struct.unpack("<I", tx[24:28])[0]
    prog = tx[28: 28 + plen]
    rest = tx[28 + plen:]

    def with_program(p, version=1):
        return struct.pack("<QQQ", version, 0, 2 ** 40) + struct.pack("<I", len(p)) + p + rest

    for bad, want in ((with_program(prog + b"\x15"), 2),                 # an instruction of the full VM (issue)
                      (with_program(prog, version=2), 2),                 # a later transaction version
                      (with_program(prog + b"\x02"), 1),                  # drop on an empty stack: invalid
                      (with_program(prog[:-5]), 1),                       # an output missing: a value left on the stack
                      (with_program(b""), 2),                             # no cloak at all: nothing for the proof system
                      (tx[:-1], 1), (tx + b"\x00", 1), (tx[:20], 1)):
        assert oracle.tx_id(bad)[0] == want
        assert prepare(host, bad)[0] == want


def _with_program(tx, prog):
    plen = struct.unpack("<I", tx[24:28])[0]
    return tx[:24] + struct.pack("<I", len(prog)) + prog + tx[28 + plen:]


def test_transaction_with_more_than_65536_hash_jobs(host, oracle):
    """hash-job slots were 16-bit and a structurally valid transaction of ~700 KB wrapped them (a heap
    overflow on the slot memory, aliased slots in the transaction ID).  17 000 empty outputs after a payment: two contract-id
    chains, a Merkle leaf and a node each -- about 68 000 jobs.  The transaction ID must be the oracle's (whose log, stack and
    key list grow with the program since this round: no capacity is a rule of the format), one at a time and in lockstep."""
    tx = payment(oracle, 1, 1, 4242)
    plen = struct.unpack("<I", tx[24:28])[0]
    prog = tx[28: 28 + plen]
    reps = 17000
    tail = b"".join(b"\x00" + struct.pack("<I", 32) + hashlib.sha256(b"pred %d" % i).digest() + b"\x1c" + struct.pack("<I", 0) for i in range(reps))
    big = _with_program(tx, prog + tail)
    assert len(big) > 600_000
    want_rc, want_id, wa, wb = oracle.tx_id(big)
    assert (want_rc, wa, wb) == (0, 1, 1)
    assert want_id != oracle.tx_id(tx)[1]
    rc, txid, a, b, com, ss, sp, po, pl = prepare(host, big)
    assert (rc, txid, a, b) == (0, want_id, 1, 1)
    assert po + pl == len(big)
    # a shorter relative of it, eight in lockstep (AVX-512 where the CPU has it) against one at a time: 9 000 outputs each,
    # the ids of different transactions in one group
    txs = []
    for q in range(8):
        t = payment(oracle, 1, 1, 4300 + q)
        pl_q = struct.unpack("<I", t[24:28])[0]
        txs.append(_with_program(t, t[28: 28 + pl_q] + tail[: 42 * 9000]))
    blob = b"".join(txs)
    offs = (C.c_uint64 * 9)()
    for i, t in enumerate(txs):
        offs[i + 1] = offs[i] + len(t)
    agg = hashlib.shake_256(b"agg").digest(32 * 8)
    got = {}
    for mode in (0, 1):
        st, ids, dig = C.create_string_buffer(8), C.create_string_buffer(32 * 8), C.create_string_buffer(64 * 8)
        host.zkhost_tx_prepare_group(blob, offs, C.c_size_t(8), mode, agg, st, ids, dig)
        got[mode] = (st.raw, ids.raw, dig.raw)
    assert got[0] == got[1] and got[0][0] == bytes(8)
    for q in (0, 5):
        assert oracle.tx_id(txs[q])[1] == got[0][1][32 * q: 32 * q + 32]


def test_long_programs_beyond_the_old_fixed_capacities_agree_with_the_oracle(host, oracle):
    """More than 256 stack items, more than 160 log entries, more than 64 payload items of a signed contract, a contract
    string duplicated and spent many times (more than 64 keys): verdict and transaction ID as the oracle's."""
    tx = payment(oracle, 2, 2, 99)
    plen = struct.unpack("<I", tx[24:28])[0]
    prog = tx[28: 28 + plen]
    push = lambda b: b"\x00" + struct.pack("<I", len(b)) + b                 # noqa: E731
    drop, inp, signtx = b"\x02", b"\x1b", b"\x20"
    dup = lambda k: b"\x03" + struct.pack("<I", k)                           # noqa: E731
    out = lambda k: b"\x1c" + struct.pack("<I", k)                           # noqa: E731
    pred = prog[5 + 32: 5 + 64]                       # the first input's key: predicates that sign must decode
    # (a) 400 strings on the stack at once, then dropped
    a = prog + push(b"x") * 400 + drop * 400
    # (b) 300 outputs of no items: 301 log entries beyond the payment's
    b = prog + (push(pred) + out(0)) * 300
    # (c) a contract with 100 data items, spent: its payload comes back onto the stack, and leaves in one output:100
    contract = hashlib.sha256(b"anchor").digest() + pred + struct.pack("<I", 100) + b"".join(b"\x00" + struct.pack("<I", 3) + b"abc" for _ in range(100))
    c = prog + push(contract) + inp + signtx + push(pred) + out(100)
    # (d) one contract string spent 70 times: 70 more keys, 70 inputs
    small = hashlib.sha256(b"anchor2").digest() + pred + struct.pack("<I", 0)
    d = prog + push(small) + dup(0) * 69 + (inp + signtx) * 70
    # (e) the same with a payload item left on the stack at the end: invalid in both
    e = prog + push(contract) + inp + signtx
    for name, p, want in (("a", a, 0), ("b", b, 0), ("c", c, 0), ("d", d, 0), ("e", e, 1)):
        t = _with_program(tx, p)
        rc, txid, wa, wb = oracle.tx_id(t)
        got = prepare(host, t)
        assert rc == want and got[0] == want, name
        if want == 0:
            assert got[1] == txid and (got[2], got[3]) == (wa, wb) == (2, 2), name


def test_product_builder_makes_the_oracle_builders_bytes(host, oracle):
    """csrc/zkvm_tx_build.hpp (what bench.py and the GPU tests make their distinct transactions with) against the oracle's
    separate builder: the same bytes for the same proof and seed, for several shapes; and the transactions of
    gpu_util.built_transactions -- distinct, 1 in 64 damaged at drawn positions -- get the verdict their construction
    says from the oracle's Tx::verify."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from gpu_util import built_transactions, load_mixed_fixture
    host.zkhost_tx_wrap_payment.restype = C.c_size_t
    fix = load_mixed_fixture()
    for (n_in, n_out), recs in sorted(fix.items()):
        for k in (0, 7):
            com, proof = recs[k]
            seed = hashlib.sha256(b"builder %d %d %d" % (n_in, n_out, k)).digest()
            want = oracle.tx_wrap_payment(n_in, n_out, com, proof, seed, mintime=5 + k, maxtime=10 ** 9)
            out = C.create_string_buffer(len(want) + 64)
            n = host.zkhost_tx_wrap_payment(C.c_size_t(n_in), C.c_size_t(n_out), com, proof, C.c_size_t(len(proof)), seed, C.c_uint64(5 + k),
                                            C.c_uint64(10 ** 9), out, C.c_size_t(len(want) + 64))
            assert n == len(want) and out.raw[:n] == want, (n_in, n_out, k)
    txs, expected = built_transactions(192, call=3, bad_every=16)
    assert len(set(txs)) == 192 and expected.count(0) == 12
    r = hashlib.shake_256(b"built").digest(64)
    for i in list(range(0, 192, 17)) + [i for i, e in enumerate(expected) if not e]:
        assert (oracle.tx_verify(txs[i], r) == 0) == bool(expected[i]), i
    assert not set(built_transactions(64, call=4, bad_every=0)[0]) & set(txs)          # another call: other transactions


def _txcall(host, txs, proof_ok, chunk, seed, fail_at=-1, threads=4):
    blob = b"".join(txs)
    offs = (C.c_uint64 * (len(txs) + 1))()
    for i, t in enumerate(txs):
        offs[i + 1] = offs[i] + len(t)
    n = len(txs)
    bm, st = C.create_string_buffer((n + 7) // 8 + 1), C.create_string_buffer(n)
    nc, ns, leaked = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    rc = host.zkhost_txcall_selftest(C.c_size_t(n), blob, offs, bytes(proof_ok), C.c_int(threads), C.c_size_t(chunk), C.c_uint32(seed),
                                     C.c_int(fail_at), bm, st, C.byref(nc), C.byref(ns), C.byref(leaked))
    return rc, [(bm.raw[i // 8] >> (i % 8)) & 1 for i in range(n)], list(st.raw), nc.value, ns.value, leaked.value


def test_scheduling_of_a_transaction_call_on_a_stand_in_device(host, oracle):
    """csrc/tx_call.hpp (what zkgpu_tx_verify_batch runs on the GPU) driven on the CPU by a stand-in device whose stages
    finish on threads of their own after random delays (hostlib.cpp: HostTxDevice -- aggregated keys and signature
    equations with the reference group arithmetic, cloak proofs by a table): one chunk (proofs first) and many chunks (keys
    first; more chunks than ring slots; several signature stages), ragged last chunks, transactions of other shapes and
    damaged ones in every part, unsupported ones -- every accept bit and status byte as constructed; then a device fault
    injected at EVERY device operation in turn: an error, all-zero outputs, no hang, every staged chunk released."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from gpu_util import built_transactions
    txs, expected = built_transactions(150, call=9, bad_every=8)
    # kinds 0 (proof) are the only ones the stand-in cannot see by itself: they go into its table
    proof_ok = [1] * len(txs)
    clean, _ = built_transactions(150, call=9, bad_every=0)
    for i, (a, b) in enumerate(zip(txs, clean)):
        plen = struct.unpack("<I", b[24:28])[0]
        if a[28 + plen + 64:] != b[28 + plen + 64:]:
            proof_ok[i] = 0
    assert 2 <= proof_ok.count(0) <= 6 and expected.count(0) == 18
    txs[5] = payment(oracle, 1, 2, 77)                                    # other shapes in between
    txs[77] = payment(oracle, 3, 2, 78, two_flavors=True)
    txs[100] = txs[100][:8] + b"\x00" * 8 + txs[100][16:]                 # (mintime 0: still valid? no -- the txid changes: rejected)
    expected[100] = 0
    t = bytearray(txs[120]); t[0] = 2; txs[120] = bytes(t)                # a later version: outside the subset
    expected[120] = 0
    want_status = [0 if e else 1 for e in expected]
    want_status[120] = 2
    for chunk, seed in ((0, 1), (8, 2), (16, 3), (24, 4), (40, 5), (64, 6), (1000, 7)):
        rc, bits_, st, nc, ns, leaked = _txcall(host, txs, proof_ok, chunk, seed)
        assert rc == 0 and leaked == 0, (chunk, rc)
        assert bits_ == expected, (chunk, [i for i in range(len(txs)) if bits_[i] != expected[i]])
        assert st == want_status, chunk
        if chunk == 8:
            assert nc == 19 and ns >= 1                                   # more chunks than ring slots (6)
    # two calls driven by ONE thread through start / step / done / finish, one stage slot each (the engine of
    # zkgpu_tx_verify_submit keeps two rounds in flight this way)
    blob = b"".join(txs)
    offs = (C.c_uint64 * (len(txs) + 1))()
    for i, t in enumerate(txs):
        offs[i + 1] = offs[i] + len(t)
    for split, chunk, seed in ((64, 0, 1), (80, 16, 2), (8, 24, 3), (144, 8, 4)):
        bm, st = C.create_string_buffer((len(txs) + 7) // 8 + 1), C.create_string_buffer(len(txs))
        leaked = C.c_size_t(0)
        rc = host.zkhost_txcall_pair_selftest(C.c_size_t(len(txs)), C.c_size_t(split), blob, offs, bytes(proof_ok), C.c_int(4), C.c_size_t(chunk),
                                              C.c_uint32(seed), bm, st, C.byref(leaked))
        assert rc == 0 and leaked.value == 0, (split, rc)
        assert [(bm.raw[i // 8] >> (i % 8)) & 1 for i in range(len(txs))] == expected, split
        assert list(st.raw) == want_status, split
    # a fault at every device operation of a many-chunk call, and of a one-chunk call
    for chunk in (16, 0):
        ops, k = 0, 0
        while True:
            rc, bits_, st, nc, ns, leaked = _txcall(host, txs[:64], proof_ok[:64], chunk, 100 + k, fail_at=k)
            if rc == 0:
                assert bits_ == expected[:64]
                break
            assert rc == -3 and not any(bits_) and 0 not in st and leaked == 0, (chunk, k)
            k += 1
            assert k < 200
        assert k >= (9 if chunk else 5), (chunk, k)                       # that many device operations could fail


def test_committed_transaction_fixture_is_what_the_oracle_accepts(host, oracle):
    """tests/golden/tx_2x2_1024_wrappers.bin (+ the committed cloak proofs): a sample verifies under the oracle, and the
    product's host half reads the same transaction IDs out of all 1024."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from gpu_util import load_tx_fixture
    txs = load_tx_fixture()
    assert len(txs) == 1024 and len(set(txs)) == 1024
    for i in (0, 1, 511, 1023):
        assert oracle.tx_verify(txs[i], hashlib.shake_256(b"fixture %d" % i).digest(64)) == 0
    for i in range(0, 1024, 37):
        rc, txid, a, b = oracle.tx_id(txs[i])
        got = prepare(host, txs[i])
        assert rc == 0 and got[0] == 0 and got[1] == txid and (got[2], got[3]) == (a, b) == (2, 2)


def test_lockstep_hashing_of_eight_transactions_equals_one_at_a_time(host, oracle):
    """merlin_x8.hpp: the payment VM's hashing with eight Keccak states per AVX-512 register against the same plans run one
    transaction at a time -- status, transaction ID, MuSig coefficients and signature terms (after the challenge) byte
    for byte: uniform groups (the committed transactions), groups with a damaged or foreign transaction in them (the
    rest still runs in lockstep), groups of other shapes side by side (one at a time), ragged last groups; and the
    one-at-a-time form is the one every other test holds against the oracle."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from gpu_util import load_tx_fixture
    rng = random.Random(808)
    base = load_tx_fixture()
    txs = list(base[:203])                                                     # ragged: 25 groups of eight and one of three
    for i in rng.sample(range(len(txs)), 30):
        t = bytearray(txs[i])
        kind = rng.randrange(4)
        if kind == 0:
            t[rng.randrange(len(t))] ^= 1 << rng.randrange(8)
        elif kind == 1:
            t[0] = 2
        elif kind == 2:
            t = t[: len(t) - 1 - rng.randrange(50)]
        else:
            t = bytearray(payment(oracle, rng.choice([1, 3]), rng.choice([1, 2]), 5000 + i))      # another shape in the group
        txs[i] = bytes(t)
    txs += [payment(oracle, 3, 2, 6000 + i, two_flavors=True) for i in range(9)]                    # a uniform run of another shape
    txs += [payment(oracle, 6, 3, 7000 + i) for i in range(8)]                                     # more keys than fit inline
    blob = b"".join(txs)
    offs = (C.c_uint64 * (len(txs) + 1))()
    for i, t in enumerate(txs):
        offs[i + 1] = offs[i] + len(t)
    agg = hashlib.shake_256(b"aggregated keys").digest(32 * len(txs))
    out = {}
    for mode in (0, 1):
        st, ids, dig = C.create_string_buffer(len(txs)), C.create_string_buffer(32 * len(txs)), C.create_string_buffer(64 * len(txs))
        have = host.zkhost_tx_prepare_group(blob, offs, C.c_size_t(len(txs)), mode, agg, st, ids, dig)
        out[mode] = (st.raw, ids.raw, dig.raw)
    assert out[0] == out[1]
    assert out[0][0].count(b"\x00") >= 190 and out[0][0].count(b"\x02") >= 3
    # the keys-first pass of zkgpu_tx_verify_batch (a plan that drops every hash job but the MuSig coefficients') leaves the
    # same status, arity, commitments, s, a_i, R and keys as the full pass -- in lockstep and one at a time
    rows = {}
    for lockstep in (0, 1):
        for keys_only in (0, 1):
            st, dig = C.create_string_buffer(len(txs)), C.create_string_buffer(64 * len(txs))
            host.zkhost_tx_rows_group(blob, offs, C.c_size_t(len(txs)), lockstep, keys_only, st, dig)
            rows[lockstep, keys_only] = (st.raw, dig.raw)
    assert rows[0, 0] == rows[0, 1] == rows[1, 0] == rows[1, 1]
    assert rows[0, 0][0] == out[0][0]
    if not have:
        pytest.skip("no AVX-512 on this CPU: both modes ran one transaction at a time")
    for i in (0, 7, 100, 202, 203, 211, 212, 219):                              # and the one-at-a-time form is the oracle's
        rc, txid, a, b = oracle.tx_id(txs[i])
        assert rc == out[1][0][i] and (rc != 0 or txid == out[1][1][32 * i: 32 * i + 32])


@pytest.mark.gpu
def test_transactions_verified_on_the_device_equal_oracle():
    """zkgpu_tx_verify_batch: a batch of serialized transactions of several shapes, some damaged in every part, some outside
    the subset -- accept bits and status bytes against the oracle's Tx::verify."""
    import oracle.binding as oracle
    from zkvm_amd import Context
    from zkvm_amd.verifier import BulletproofGens, BlockVerifier
    ctx = Context(0)
    gens = BulletproofGens(ctx, 256, table_bits=8)
    bv = BlockVerifier(ctx, gens)
    try:
        txs = []
        for i in range(40):
            shape = [(1, 1), (2, 2), (3, 2), (1, 2), (2, 1)][i % 5]
            txs.append(payment(oracle, shape[0], shape[1], 1000 + i, two_flavors=(i % 3 == 0) and min(shape) >= 2))
        for i, at in ((3, 8), (7, 40), (11, -1), (13, None), (17, "s"), (19, "R"), (23, "v2"), (29, "op")):
            t = bytearray(txs[i])
            plen = struct.unpack("<I", t[24:28])[0]
            if at == "s":
                t[28 + plen + 40] ^= 4
            elif at == "R":
                t[28 + plen + 2] ^= 4
            elif at == "v2":
                t[0] = 2
            elif at == "op":
                t = bytearray(struct.pack("<QQQ", 1, 0, 2 ** 40) + struct.pack("<I", plen + 1) + bytes(t[28: 28 + plen]) + b"\x15" + bytes(t[28 + plen:]))
            elif at is None:
                t = t[:-7]
            else:
                t[at] ^= 1
            txs[i] = bytes(t)
        r = hashlib.shake_256(b"tx device r").digest(64)
        want = [oracle.tx_verify(t, r) for t in txs]
        assert want.count(0) == 32 and want.count(2) == 2
        # inert until a format is named: everything "outside the subset", nothing accepted
        bm0, st0 = bv.verify_txs(txs, host_threads=4)
        assert bm0 == bytes(len(bm0)) and list(st0) == [2] * len(txs)
        bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
        bm, st = bv.verify_txs(txs, host_threads=4)
        assert list(st) == want
        assert [(bm[i // 8] >> (i % 8)) & 1 for i in range(len(txs))] == [1 if w == 0 else 0 for w in want]
    finally:
        bv.close()
        gens.close()
        ctx.close()


@pytest.mark.gpu
def test_the_1024_fixture_transactions_on_the_device():
    """the benched input of bench.py's tx_verify leg: all 1024 committed transactions accepted, and a damaged copy of
    every 16th rejected, alone"""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from gpu_util import load_tx_fixture
    from zkvm_amd import Context
    from zkvm_amd.verifier import BulletproofGens, BlockVerifier
    txs = load_tx_fixture()
    for i in range(5, 1024, 16):
        t = bytearray(txs[i])
        t[(97 * i) % len(t)] ^= 1 << (i % 8)
        txs[i] = bytes(t)
    ctx = Context(0)
    gens = BulletproofGens(ctx, 256, table_bits=12)
    bv = BlockVerifier(ctx, gens)
    bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
    try:
        bm, st = bv.verify_txs(txs)
        with pytest.raises(ValueError):                        # lengths that do not add up to the buffer: refused before the C call
            bv.verify_txs_packed(b"".join(txs), [len(t) for t in txs[:-1]])
        assert bv.verify_txs_packed(b"".join(txs), [len(t) for t in txs]) == (bm, st)
        bad = set(range(5, 1024, 16))
        assert [i for i in range(1024) if not (bm[i // 8] >> (i % 8)) & 1] == sorted(bad)
        assert all((st[i] != 0) == (i in bad) for i in range(1024))
    finally:
        bv.close()
        gens.close()
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_tx,tx_chunk,kept", [(7168, 0, None), (7168, 1000, None), (7168, 2304, 2051), (12288, 1000, None)])
def test_a_long_call_travels_through_the_stages_in_chunks(n_tx, tx_chunk, kept):
    """zkgpu_tx_verify_batch on 7168 (12 288) transactions in ONE call: the call is cut into chunks whose host stages (VM, signature
    transcripts) run beside the device stages of the others (aggregated keys, signature equations, cloak proofs on the
    lanes) -- with the default chunking (one chunk up to 8192 transactions), with chunks of 1000 (eight chunks: the staging
    ring of six is reused), of 2304 (four chunks, the last one short), and 12 288 transactions in chunks of 1000 (two runs of
    chunks: three signature stages, the key stages of thirteen chunks in turn on two contexts); once with a verifier that
    keeps only 2051 transactions' VM results between calls (the rest of the call's live in memory of the call's own).  Sixty transactions are damaged in every part, in every chunk: their verdicts are the
    oracle's Tx::verify, everybody else's is "accepted"; and status 0 appears exactly beside accept bits of 1."""
    import random
    import sys
    import oracle.binding as oracle
    sys.path.insert(0, os.path.dirname(__file__))
    from gpu_util import load_tx_fixture
    from zkvm_amd import Context
    from zkvm_amd.verifier import BulletproofGens, BlockVerifier
    base = load_tx_fixture()
    txs = [base[i % 1024] for i in range(n_tx)]
    rng = random.Random(77 + tx_chunk + n_tx)
    damaged = {}
    for i in sorted(rng.sample(range(len(txs)), 60)):
        t = bytearray(txs[i])
        plen = struct.unpack("<I", t[24:28])[0]
        kind = len(damaged) % 6
        if kind == 0:
            t[28 + plen + 32 + rng.randrange(31)] ^= 1 << rng.randrange(8)      # signature scalar s
        elif kind == 1:
            t[28 + plen + rng.randrange(32)] ^= 1 << rng.randrange(8)           # signature point R
        elif kind == 2:
            t[28 + rng.randrange(plen)] ^= 1 << rng.randrange(8)                # somewhere in the program
        elif kind == 3:
            t[len(t) - 1 - rng.randrange(900)] ^= 1 << rng.randrange(8)         # somewhere in the proof
        elif kind == 4:
            t[0] = 2                                                            # another version: outside the subset
        else:
            t = t[: len(t) - 1 - rng.randrange(40)]                             # truncated
        txs[i] = bytes(t)
        damaged[i] = None
    r = hashlib.shake_256(b"long call").digest(64)
    for i in damaged:
        damaged[i] = oracle.tx_verify(txs[i], r)
    assert set(damaged.values()) == {1, 2} or set(damaged.values()) == {0, 1, 2}       # (a flipped bit may land in dead bytes)
    ctx = Context(0)
    gens = BulletproofGens(ctx, 256, table_bits=12)
    bv = BlockVerifier(ctx, gens)
    bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
    bv.set_tx_chunk(tx_chunk)
    if kept is not None:
        bv.set_tx_statements_kept(kept)
    try:
        for _ in range(2):
            bm, st = bv.verify_txs(txs, host_threads=8)
            want_status = [damaged.get(i, 0) for i in range(len(txs))]
            assert list(st) == want_status
            assert [(bm[i // 8] >> (i % 8)) & 1 for i in range(len(txs))] == [1 if w == 0 else 0 for w in want_status]
    finally:
        bv.close()
        gens.close()
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("engine_rounds", ["1", "2"])
def test_transaction_calls_in_flight_on_one_verifier_keep_their_own_verdicts(engine_rounds, monkeypatch):
    """zkgpu_tx_verify_submit / _wait: calls in flight on ONE verifier, merged by its engine into rounds
    -- one round at a time, and two in flight (ZKGPU_TX_ROUNDS, read when the verifier's engine starts; the library's own
    choice depends on the process's hardware queues: DESIGN.md sec 10.3).
    DISTINCT transactions (gpu_util.built_transactions -- the inputs of bench.py's tx_verify leg: one in 16 damaged at drawn
    positions here), their constructed expectation held against the oracle's Tx::verify for every damaged one and a sample
    of the others; 12 calls of 300 submitted from four threads at once, waited for in another order; a synchronous call and
    a block of cloak proofs between them; a transaction outside the subset and one of another shape keep their own status;
    the engine has run fewer rounds than calls (they were merged)."""
    import sys
    import threading
    import oracle.binding as oracle
    sys.path.insert(0, os.path.dirname(__file__))
    from gpu_util import bits, built_transactions, load_cloak_fixture
    from zkvm_amd import Context
    from zkvm_amd.verifier import BulletproofGens, BlockVerifier, CloakTx
    monkeypatch.setenv("ZKGPU_TX_ROUNDS", engine_rounds)
    txs, expected = built_transactions(3600, call=21, bad_every=16)
    assert len(set(txs)) == 3600
    r = hashlib.shake_256(b"in flight").digest(64)
    for i in [i for i, e in enumerate(expected) if not e][:40] + list(range(0, 3600, 211)):
        assert (oracle.tx_verify(txs[i], r) == 0) == bool(expected[i]), i
    status = [0 if e else 1 for e in expected]
    t = bytearray(txs[700]); t[0] = 2; txs[700] = bytes(t); expected[700] = 0; status[700] = 2           # a later version
    txs[1300] = payment(oracle, 1, 2, 501); expected[1300] = 1; status[1300] = 0                            # another shape in the middle
    ctx = Context(0)
    gens = BulletproofGens(ctx, 256, table_bits=10)
    bv = BlockVerifier(ctx, gens)
    try:
        cid = bv.submit_txs(txs[:10])                               # no format named yet: everything is outside the subset
        bm, st = bv.wait_txs(cid)
        assert bm == bytes(2) and list(st) == [2] * 10
        bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
        ids = [None] * 12
        def submit(k):
            ids[k] = bv.submit_txs(txs[300 * k: 300 * (k + 1)], host_threads=8)
        th = [threading.Thread(target=lambda a=a: [submit(k) for k in range(a, 12, 4)]) for a in range(4)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        # a synchronous call and a block of proofs while the calls are in flight: they take their turn
        fix, n_in, n_out, plen = load_cloak_fixture()
        assert bits(bv.verify([CloakTx(n_in, n_out, *fix[i]) for i in range(50)], hashlib.shake_256(b"blk").digest(64 * 50)), 50) == [1] * 50
        bm, st = bv.verify_txs(txs[:64], host_threads=8)
        assert bits(bm, 64) == expected[:64]
        for k in (11, 3, 0, 7, 1, 2, 4, 5, 6, 8, 9, 10):
            bm, st = bv.wait_txs(ids[k])
            assert bits(bm, 300) == expected[300 * k: 300 * (k + 1)], k
            assert list(st) == status[300 * k: 300 * (k + 1)], k
        rounds, calls = bv.tx_stats()
        assert calls == 13 and rounds < calls
        with pytest.raises(Exception):
            bv.lib.zkgpu_tx_verify_wait.restype = C.c_int
            rc = bv.lib.zkgpu_tx_verify_wait(bv.h, ids[0], C.create_string_buffer(64), None)
            assert rc == 0                                              # an id is waited for once: this is ZKGPU_EINVAL
    finally:
        bv.close()
        gens.close()
        ctx.close()


def test_random_programs_get_the_same_verdict_from_both_implementations(host, oracle):
    """Differential fuzzing of the two independently written VMs (zkvm_tx.hpp, oracle/zkvm_tx.c): random instruction
    sequences over the subset's opcodes (and a few outside it), random immediates, spliced fragments of a valid program --
    status and, when the program runs, transaction ID and cloak shape must agree."""
    rng = random.Random(20240)
    base = payment(oracle, 2, 2, 4242)
    plen = struct.unpack("<I", base[24:28])[0]
    prog, rest = base[28: 28 + plen], base[28 + plen:]
    contract = prog[5: 5 + 133]                                   # the first pushed contract
    com = [prog[o: o + 32] for o in range(5 + 133 + 2 + 5 + 133 + 2 + 5, 5 + 133 + 2 + 5 + 133 + 2 + 5 + 4 * 38, 38)]

    def u32(x):
        return struct.pack("<I", x)

    def fragment():
        k = rng.randrange(14)
        if k == 0: return b"\x00" + u32(len(contract)) + contract + b"\x1b"
        if k == 1: return b"\x20"
        if k == 2: return b"\x00" + u32(32) + rng.choice(com) + b"\x06"
        if k == 3: return b"\x18" + u32(rng.choice([1, 2, 2, 3])) + u32(rng.choice([1, 2, 2, 65]))
        if k == 4: return b"\x00" + u32(32) + bytes(rng.getrandbits(8) for _ in range(32)) + b"\x1c" + u32(rng.choice([0, 1, 1, 2]))
        if k == 5: return b"\x02"
        if k == 6: return b"\x03" + u32(rng.randrange(4))
        if k == 7: return b"\x04" + u32(rng.randrange(4))
        if k == 8: return b"\x00" + u32(rng.choice([0, 1, 31, 32, 33, 2 ** 31])) + bytes(rng.getrandbits(8) for _ in range(rng.randrange(40)))
        if k == 9: return bytes([rng.choice([0x01, 0x05, 0x15, 0x1f, 0x21, 0xff])])
        if k == 10: return b"\x1b"
        if k == 11: return b"\x06"
        if k == 12:                                                # a damaged contract
            c = bytearray(contract); c[rng.randrange(64, len(c))] ^= 1 << rng.randrange(8)
            return b"\x00" + u32(len(c)) + bytes(c) + b"\x1b"
        a = rng.randrange(len(prog)); return prog[a: a + rng.randrange(1, 200)]

    seen = {0: 0, 1: 0, 2: 0}
    for case in range(6000):
        if case % 5 == 0:                                          # mutate the valid program
            p = bytearray(prog)
            for _ in range(rng.randrange(1, 4)):
                at = rng.randrange(len(p))
                p[at: at + rng.randrange(0, 3)] = fragment() if rng.random() < 0.5 else bytes([rng.getrandbits(8)])
            p = bytes(p)
        else:
            p = b"".join(fragment() for _ in range(rng.randrange(1, 12)))
        tx = struct.pack("<QQQ", rng.choice([1, 1, 1, 2]), 5, rng.choice([5, 9, 4])) + u32(len(p)) + p + rest
        want = oracle.tx_id(tx)
        got = prepare(host, tx)
        if (got[0], want[0]) == (1, 0):
            # the product's preparation also decodes the signing keys (it aggregates them); the oracle does that when it
            # verifies: a key that is no point is "invalid" on both sides, just at different stages
            assert oracle.tx_verify(tx, bytes(64)) == 1, (case, p.hex())
            seen[1] += 1
            continue
        assert got[0] == want[0], (case, p.hex())
        seen[want[0]] += 1
        if want[0] == 0:
            assert got[1] == want[1] and (got[2], got[3]) == (want[2], want[3]), (case, p.hex())
    assert seen[1] > 1200 and seen[2] > 400 and seen[0] > 20, seen
