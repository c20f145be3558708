"""Product host logic on CPU (no GPU): scalar field, Merlin transcript and the R1CS verifier's scalar
preparation (zkvm_amd/csrc/{scalar,merlin,r1cs_verifier}.hpp via libzkhost.so) vs the oracle.
The MSM terms must agree byte for byte: same challenges, same flattened constraints, same s vector."""
import ctypes as C
import hashlib
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = 2**252 + 27742317777372353535851937790883648493


@pytest.fixture(scope="module")
def host():
    from zkvm_amd import build
    build.build()
    return C.CDLL(os.path.join(ROOT, "zkvm_amd", "lib", "libzkhost.so"))


def _op(host, op, a, b=0):
    out = C.create_string_buffer(32)
    assert host.zkhost_scalar_op(op, a.to_bytes(64, "little"), b.to_bytes(64, "little"), out) == 0
    return int.from_bytes(out.raw, "little")


def test_scalar_field_vs_bigint(host):
    rng = random.Random(9)
    edge = [0, 1, L - 1, L, L + 1, 2**252, 2**256 - 1, 2**512 - 1, L * L, 2**511 + 7]
    xs = edge + [rng.getrandbits(512) for _ in range(150)]
    for a in xs:
        assert _op(host, 5, a) == a % L
        assert _op(host, 3, a) == (-a) % L
        for b in rng.sample(xs, 4):
            assert _op(host, 0, a, b) == (a + b) % L
            assert _op(host, 1, a, b) == (a - b) % L
            assert _op(host, 2, a, b) == (a * b) % L
    for a in xs[:6] + xs[10:14]:
        if a % L:
            assert _op(host, 4, a) == pow(a, -1, L)
    assert host.zkhost_scalar_is_canonical((L - 1).to_bytes(32, "little")) == 1
    assert host.zkhost_scalar_is_canonical(L.to_bytes(32, "little")) == 0
    assert host.zkhost_scalar_is_canonical((2**256 - 1).to_bytes(32, "little")) == 0


def _merlin(host, label, msgs, ch_label, n):
    k = len(msgs)
    labels = (C.c_char_p * k)(*[m[0] for m in msgs])
    bufs = [C.create_string_buffer(m[1], len(m[1])) for m in msgs]
    ptrs = (C.c_void_p * k)(*[C.addressof(b) for b in bufs])
    lens = (C.c_size_t * k)(*[len(m[1]) for m in msgs])
    out = C.create_string_buffer(n)
    host.zkhost_merlin(label, k, labels, ptrs, lens, ch_label, out, C.c_size_t(n))
    return out.raw


def test_merlin_known_answer_and_vs_oracle(host, oracle):
    got = _merlin(host, b"test protocol", [(b"some label", b"some data")], b"challenge", 32)
    assert got.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    rng = random.Random(10)
    msgs = [(b"l%d" % i, bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 1, 32, 165, 166, 167, 500])))) for i in range(12)]
    t = oracle.MerlinTranscript(b"ZkVM.r1cs")
    for lab, m in msgs:
        t.append_message(lab, m)
    assert _merlin(host, b"ZkVM.r1cs", msgs, b"c", 200) == t.challenge_bytes(b"c", 200)


def _prepare(host, com, n_in, n_out, proof, r, cap):
    ds, dp = C.create_string_buffer(32 * 128), C.create_string_buffer(32 * 128)
    ss, si = C.create_string_buffer(32 * (2 + 2 * cap)), (C.c_uint32 * (2 + 2 * cap))()
    nd, ns, pn = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    rc = host.zkhost_cloak_prepare(com, C.c_size_t(n_in), C.c_size_t(n_out), proof, C.c_size_t(len(proof)), r,
                                   C.c_size_t(cap), ds, dp, C.byref(nd), ss, si, C.byref(ns), C.byref(pn))
    if rc:
        return None
    return ds.raw[: 32 * nd.value], dp.raw[: 32 * nd.value], ss.raw[: 32 * ns.value], list(si[: ns.value]), pn.value


@pytest.mark.parametrize("n_in,n_out", [(1, 1), (2, 2), (1, 2), (3, 3)])
def test_verifier_msm_terms_equal_oracle(host, oracle, n_in, n_out):
    com, proofs = oracle.cloak_prove_batch(3, n_in, n_out, bytes([n_in * 16 + n_out] * 32), threads=2)
    w = 64 * (n_in + n_out)
    for i, proof in enumerate(proofs):
        r = hashlib.shake_256(b"r%d" % i).digest(64)
        want = oracle.cloak_verify_prepare(com[w * i: w * (i + 1)], n_in, n_out, proof, r)
        cap = 256
        got = _prepare(host, com[w * i: w * (i + 1)], n_in, n_out, proof, r, cap)
        assert got is not None and want is not None
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2] and got[4] == want[3]
        pn = got[4]
        assert got[3] == [0, 1] + [2 + j for j in range(pn)] + [2 + cap + j for j in range(pn)]


def test_verifier_prepare_rejects_malformed(host, oracle):
    com, proofs = oracle.cloak_prove_batch(1, 2, 2, b"\x09" * 32)
    r = bytes(64)
    proof = proofs[0]
    assert _prepare(host, com, 2, 2, proof, r, 256) is not None
    assert _prepare(host, com, 2, 2, proof, r, 128) is None                        # not enough generators
    assert _prepare(host, com, 2, 2, proof[:-32], r, 256) is None
    assert _prepare(host, com, 2, 2, b"\x00" + proof[1:], r, 256) is None          # wrong version tag
    bad = bytearray(proof); bad[1 + 32 * 7: 1 + 32 * 8] = bytes(32)
    assert _prepare(host, com, 2, 2, bytes(bad), r, 256) is None                   # T_3 = identity
    bad = bytearray(proof); bad[-32:] = L.to_bytes(32, "little")
    assert _prepare(host, com, 2, 2, bytes(bad), r, 256) is None                   # b not canonical
    assert _prepare(host, com, 1, 1, proof, r, 256) is None                        # k does not match the statement


def compact_proof(proof: bytes) -> bytes:
    """the one-phase wire format of a proof whose A_I2, A_O2, S2 are the identity: version byte 0, the three points left out"""
    assert proof[0] == 1 and proof[97:193] == bytes(96)
    return b"\x00" + proof[1:97] + proof[193:]


def test_one_phase_wire_format_equals_two_phase_with_identities(host, oracle):
    """upstream R1CSProof::to_bytes writes 13 + 2k elements under version byte 0 when the statement has no second phase
    (from_bytes restores the identity for A_I2, A_O2, S2): oracle and host verifier produce the same multiscalar
    multiplication from either form (1x1 cloak: one phase), and reject a version byte that disagrees with the length."""
    com, proofs = oracle.cloak_prove_batch(2, 1, 1, b"\x21" * 32)
    for i, proof in enumerate(proofs):
        c = com[128 * i: 128 * (i + 1)]
        r = hashlib.shake_256(b"one phase %d" % i).digest(64)
        short = compact_proof(proof)
        assert len(short) == len(proof) - 96
        want = oracle.cloak_verify_prepare(c, 1, 1, proof, r)
        assert want is not None and oracle.cloak_verify_prepare(c, 1, 1, short, r) == want
        assert oracle.cloak_verify(c, 1, 1, short, r)
        assert _prepare(host, c, 1, 1, short, r, 64) == _prepare(host, c, 1, 1, proof, r, 64) != None
        assert _prepare(host, c, 1, 1, b"\x01" + short[1:], r, 64) is None          # two-phase tag on the short form
        assert _prepare(host, c, 1, 1, b"\x00" + proof[1:], r, 64) is None          # one-phase tag on the long form
        assert oracle.cloak_verify_prepare(c, 1, 1, b"\x00" + proof[1:], r) is None
        bad = bytearray(short); bad[40] ^= 1
        assert not oracle.cloak_verify(c, 1, 1, bytes(bad), r)
    # a two-phase statement sent in the short form is a proof with identities where points were: parsed, then rejected
    com2, proofs2 = oracle.cloak_prove_batch(1, 2, 2, b"\x22" * 32)
    short2 = b"\x00" + proofs2[0][1:97] + proofs2[0][193:]
    assert not oracle.cloak_verify(com2, 2, 2, short2, bytes(64))


def _golden_cloak():
    import struct
    raw = open(os.path.join(ROOT, "tests", "golden", "cloak_2x2_proofs.bin"), "rb").read()
    assert raw[:8] == b"ZKCLOAK1"
    count, n_in, n_out, plen = struct.unpack("<IIII", raw[8:24])
    w = 64 * (n_in + n_out)
    body = raw[24:]
    return [(body[(w + plen) * i: (w + plen) * i + w], body[(w + plen) * i + w: (w + plen) * (i + 1)]) for i in range(count)]


def test_transcript_tape_equals_transcript_class(host):
    """The per-shape STROBE tape that k_transcript runs on the device (transcript_tape.hpp: constant
    bytes merged per word, data runs, permutation points, challenge slots) replayed by the host
    interpreter yields the challenges of the Transcript class on the same proof bytes."""
    txs = _golden_cloak()
    for i in (0, 7, 63):
        com, proof = txs[i]
        a, b = C.create_string_buffer(32 * 64), C.create_string_buffer(32 * 64)
        n = host.zkhost_tape_challenges(2, 2, com, proof, C.c_size_t(len(proof)), a, b, C.c_size_t(64))
        assert n == 5 + 8 + 8
        assert a.raw[: 32 * n] == b.raw[: 32 * n]
        assert len({a.raw[32 * j: 32 * j + 32] for j in range(n)}) == n


def _gens_bytes(oracle, cap):
    B, Bb = oracle.pedersen_gens()
    return B + Bb + b"".join(oracle.bulletproof_gens(cap, "G")) + b"".join(oracle.bulletproof_gens(cap, "H"))


@pytest.mark.parametrize("shape,cap", [((1, 1), 64), ((2, 2), 256), ((1, 2), 256), ((3, 2), 256)])
def test_product_prover_equals_oracle_prover(host, oracle, shape, cap):
    """r1cs_prover.hpp (phased prover, MSM rows over the generator set; here evaluated by the test
    library's reference MSM) against the oracle's prover on the same witness and seed: commitments and
    proof are byte-identical (same transcript, same TranscriptRng draws, same gadget witness), and the
    oracle's verifier accepts.  Covers merge + split (one flavor), pass-through (two flavors), padding."""
    n_in, n_out = shape
    rng = random.Random(1000 * n_in + n_out)
    gens = _gens_bytes(oracle, cap)
    for case in range(2):
        fl = [rng.randrange(2**250).to_bytes(32, "little") for _ in range(2)]
        two = case == 1 and n_in >= 2 and n_out >= 2
        q_in = [rng.randrange(2**40) for _ in range(n_in)]
        f_in = [fl[j & 1] if two else fl[0] for j in range(n_in)]
        tot = [0, 0]
        for a, f in zip(q_in, f_in):
            tot[fl.index(f)] += a
        q_out, f_out = [], []
        for j in range(n_out):
            fi = (j & 1) if two else 0
            last = all(((jj & 1) if two else 0) != fi for jj in range(j + 1, n_out))
            a = tot[fi] if last else tot[fi] // 3
            tot[fi] -= a
            q_out.append(a)
            f_out.append(fl[fi])
        assert tot == [0, 0]
        q, f = q_in + q_out, f_in + f_out
        seed = hashlib.sha256(b"prover %d %d %d" % (n_in, n_out, case)).digest()
        rc, want_com, want_proof, want_n = oracle.cloak_prove(q, f, n_in, n_out, seed)
        assert rc == 0
        qa = (C.c_uint64 * len(q))(*q)
        com = C.create_string_buffer(64 * len(q))
        proof = C.create_string_buffer(4096)
        plen, nm = C.c_size_t(0), C.c_size_t(0)
        rc = host.zkhost_cloak_prove(n_in, n_out, qa, b"".join(f), seed, gens, C.c_size_t(cap), com, proof,
                                     C.c_size_t(4096), C.byref(plen), C.byref(nm))
        assert rc == 0
        assert nm.value == want_n
        assert com.raw == want_com
        assert proof.raw[: plen.value] == want_proof
        assert oracle.cloak_verify(com.raw, n_in, n_out, proof.raw[: plen.value], bytes(range(64)))
        # the DEVICE prover's phase functions (prover_dev.hpp: transcript, TranscriptRng, witness from the traced
        # description, flattening, polynomials -- what the k_pv_* kernels run, one workgroup per proof) on the host
        com2 = C.create_string_buffer(64 * len(q))
        proof2 = C.create_string_buffer(4096)
        plen2 = C.c_size_t(0)
        # ... in one call per phase, and stage by stage in the device's order (one-thread stages on the lane kernels' terms:
        # the fixed-chain inverse), as k_pv_lanes / k_pv_wg run them
        for staged in (0, 1):
            host.zkhost_set_pv_staged(staged)
            try:
                rc = host.zkhost_prove_dev_cloak(n_in, n_out, qa, b"".join(f), seed, gens, C.c_size_t(cap), com2, proof2,
                                                 C.c_size_t(4096), C.byref(plen2))
            finally:
                host.zkhost_set_pv_staged(0)
            assert rc == 0
            assert com2.raw == want_com
            assert proof2.raw[: plen2.value] == want_proof, staged


def test_lazy_scalar_form_equals_canonical(host):
    """sc_dev.hpp: the lazy ten-limb form k_prepare computes in (scl: products without packing or conditional
    subtraction, limb-wise sums and differences, one exact reduction at the end) against the canonical Montgomery
    form on random chains of operations, including the edge values 0 and l - 1."""
    host.zkhost_scl_selftest.restype = C.c_uint64
    host.zkhost_scl_selftest.argtypes = [C.c_uint64, C.c_uint32]
    for seed in (1, 0x5a6b564d, 2 ** 63 + 12345):
        assert host.zkhost_scl_selftest(seed, 1500) == 0


def test_cooperative_transcript_rng_draws_equal_the_serial_generator(host):
    """PvRngCoop (prover_dev.hpp: the TranscriptRng's draws on a Keccak state spread over a wavefront -- k_pv_rng_coop --
    here through the emulated cross-lane primitives) against PvRng (one lane, word-wise), which the prover tests tie to
    the byte-wise STROBE of the oracle: same scalars, same final state, starting right after the keying (position 32)."""
    host.zkhost_rng_coop_selftest.restype = C.c_uint64
    host.zkhost_rng_coop_selftest.argtypes = [C.c_uint64, C.c_uint32]
    for seed, n in ((1, 3), (0x5a6b564d, 275), (2 ** 63 + 9, 1027)):
        assert host.zkhost_rng_coop_selftest(seed, n) == 0


def test_cooperative_keccak_emulation_equals_keccak(host, oracle):
    """keccak_coop.hpp -- one Keccak state spread over a wavefront (DPP row shifts, row swaps, ds_bpermute),
    run on the host with emulated cross-lane primitives -- against the oracle's Keccak-f[1600]."""
    lib = oracle.load()
    rng = random.Random(11)
    for it in range(40):
        st = [0] * 25 if it == 0 else [rng.getrandbits(64) for _ in range(25)]
        a = (C.c_uint64 * 25)(*st)
        b = (C.c_uint64 * 25)(*st)
        host.zkhost_keccak_coop(a)
        lib.keccak_f1600(b)
        assert list(a) == list(b), it


@pytest.mark.parametrize("shape", [(1, 1), (1, 2), (2, 2), (3, 3), (4, 4)])
def test_cooperative_transcript_segments_equal_tape(host, shape):
    """The tape regrouped into segments for k_transcript_coop (constants + byte map per permutation, challenges
    leaving at segment starts), run through the emulated wavefront: same challenges as the tape interpreter and
    the Transcript class, for every shape of the mixed fixture."""
    import struct
    raw = open(os.path.join(ROOT, "tests", "golden", "cloak_mixed.bin"), "rb").read()
    pos, fix = 12, {}
    for _ in range(struct.unpack("<I", raw[8:12])[0]):
        count, n_in, n_out, plen = struct.unpack("<IIII", raw[pos:pos + 16]); pos += 16
        w = 64 * (n_in + n_out)
        fix[(n_in, n_out)] = [(raw[pos + (w + plen) * i: pos + (w + plen) * i + w], raw[pos + (w + plen) * i + w: pos + (w + plen) * (i + 1)])
                              for i in range(count)]
        pos += (w + plen) * count
    n_in, n_out = shape
    for com, proof in fix[shape][:3]:
        a, b, c = (C.create_string_buffer(32 * 64) for _ in range(3))
        n = host.zkhost_tape_challenges(n_in, n_out, com, proof, C.c_size_t(len(proof)), a, b, C.c_size_t(64))
        n2 = host.zkhost_coop_challenges(n_in, n_out, com, proof, C.c_size_t(len(proof)), c, C.c_size_t(64))
        assert n == n2 > 5
        assert a.raw[: 32 * n] == b.raw[: 32 * n] == c.raw[: 32 * n]


@pytest.mark.parametrize("kind,param", [(1, 8), (1, 64), (2, 1), (2, 2), (2, 5)])
def test_described_constraint_system_equals_oracle_gadget(host, oracle, kind, param):
    """The generic entry point's semantics on the host: a constraint system written down as data in the test
    (tests/gpu_util.py: a range proof, a scalar shuffle -- statements the library has no built-in code for), run
    through the product's host verifier (prepare_desc), gives byte for byte the multiscalar-multiplication terms
    of the oracle's verifier for the same statement built with its own gadget code (oracle/gadgets.c)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from gpu_util import GADGET_LABEL, describe_range, describe_shuffle
    rng = random.Random(100 * kind + param)
    m, n1, n, labels, cons = describe_range(param) if kind == 1 else describe_shuffle(param)
    if kind == 1:
        values = [rng.randrange(1 << param)]
    else:
        xs = [rng.randrange(L) for _ in range(param)]
        ys = xs[:]
        rng.shuffle(ys)
        values = xs + ys
    rc, com, proof = oracle.gadget_prove(kind, param, values, bytes([kind, param]) * 16)
    assert rc == 0
    r = hashlib.shake_256(b"desc %d %d" % (kind, param)).digest(64)
    assert oracle.gadget_verify(kind, param, com, proof, r)
    want = oracle.gadget_verify_prepare(kind, param, com, proof, r)
    offs, kinds, idx, coeff, chal, power = [0], [], [], b"", [], []
    for con in cons:
        for (k_, i_, c_, ch_, pw_) in con:
            kinds.append(k_); idx.append(i_); coeff += (c_ % L).to_bytes(32, "little"); chal.append(ch_); power.append(pw_)
        offs.append(len(kinds))
    nt = max(len(kinds), 1)
    cap = 256
    ds, dp = C.create_string_buffer(32 * 128), C.create_string_buffer(32 * 128)
    ss, si = C.create_string_buffer(32 * (2 + 2 * cap)), (C.c_uint32 * (2 + 2 * cap))()
    nd, ns, pn = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
    lab = (C.c_char_p * max(len(labels), 1))(*labels)
    rc = host.zkhost_r1cs_prepare(GADGET_LABEL, m, n1, n, len(labels), lab, len(cons), (C.c_uint64 * len(offs))(*offs),
                                  (C.c_uint8 * nt)(*kinds), (C.c_uint32 * nt)(*idx), coeff, (C.c_int32 * nt)(*chal),
                                  (C.c_uint32 * nt)(*power), com, proof, C.c_size_t(len(proof)), r, C.c_size_t(cap), ds, dp,
                                  C.byref(nd), ss, si, C.byref(ns), C.byref(pn))
    assert rc == 0
    assert pn.value == want[3]
    assert ds.raw[: 32 * nd.value] == want[0] and dp.raw[: 32 * nd.value] == want[1] and ss.raw[: 32 * ns.value] == want[2]
    # a mutated proof is malformed or yields different terms; a mutated description (one coefficient) yields different terms
    bad = bytearray(proof); bad[1 + 32 * 11] ^= 1
    rc2 = host.zkhost_r1cs_prepare(GADGET_LABEL, m, n1, n, len(labels), lab, len(cons), (C.c_uint64 * len(offs))(*offs),
                                   (C.c_uint8 * nt)(*kinds), (C.c_uint32 * nt)(*idx), coeff, (C.c_int32 * nt)(*chal),
                                   (C.c_uint32 * nt)(*power), com, bytes(bad), C.c_size_t(len(proof)), r, C.c_size_t(cap), ds, dp,
                                   C.byref(nd), ss, si, C.byref(ns), C.byref(pn))
    assert rc2 != 0 or ss.raw[: 32 * ns.value] != want[2]


@pytest.mark.parametrize("kind,param", [(1, 8), (1, 64), (2, 2), (2, 5), (3, 2), (3, 8)])
def test_described_prover_equals_oracle_gadget_prover(host, oracle, kind, param):
    """desc_prover (r1cs_prover.hpp): the prover for a constraint system described as data -- witness = committed values
    + given multiplier assignments or defining constraints -- produces byte for byte the commitments and the proof of
    the oracle's prover for the same statement (its own gadget code) on the same witness and seed; kind 3 with 8 values is
    the 1032-constraint program of BASELINE.json configs[4]."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from gpu_util import GADGET_LABEL, describe_range, describe_ranges, describe_shuffle, gadget_witness
    rng = random.Random(7 * kind + param)
    m, n1, n, labels, cons = describe_range(param) if kind == 1 else describe_shuffle(param) if kind == 2 else describe_ranges(param)
    if kind == 1:
        values = [rng.randrange(1 << param)]
    elif kind == 3:
        values = [rng.randrange(1 << 64) for _ in range(param)]
    else:
        xs = [rng.randrange(L) for _ in range(param)]
        ys = xs[:]
        rng.shuffle(ys)
        values = xs + ys
    seed = hashlib.sha256(b"desc prover %d %d" % (kind, param)).digest()
    rc, want_com, want_proof = oracle.gadget_prove(kind, param, values, seed)
    assert rc == 0
    cap = 1
    while cap < max(n, 1):
        cap *= 2
    gens = oracle.pedersen_gens()
    gens = gens[0] + gens[1] + b"".join(oracle.bulletproof_gens(cap, "G")) + b"".join(oracle.bulletproof_gens(cap, "H"))
    mult_def, given = gadget_witness(kind, param, values)
    offs, kinds, idx, coeff, chal, power = [0], [], [], b"", [], []
    for con in cons:
        for (k_, i_, c_, ch_, pw_) in con:
            kinds.append(k_); idx.append(i_); coeff += (c_ % L).to_bytes(32, "little"); chal.append(ch_); power.append(pw_)
        offs.append(len(kinds))
    nt = max(len(kinds), 1)
    lab = (C.c_char_p * max(len(labels), 1))(*labels)
    com = C.create_string_buffer(32 * m)
    proof = C.create_string_buffer(4096)
    plen = C.c_size_t(0)
    gv = b"".join(a.to_bytes(32, "little") + b.to_bytes(32, "little") for a, b in given)
    rc = host.zkhost_r1cs_prove(GADGET_LABEL, m, n1, n, len(labels), lab, len(cons), (C.c_uint64 * len(offs))(*offs),
                                (C.c_uint8 * nt)(*kinds), (C.c_uint32 * nt)(*idx), coeff, (C.c_int32 * nt)(*chal),
                                (C.c_uint32 * nt)(*power), (C.c_uint32 * max(len(mult_def), 1))(*mult_def),
                                b"".join(v.to_bytes(32, "little") for v in values), gv, C.c_size_t(len(given)), seed, gens,
                                C.c_size_t(cap), com, proof, C.c_size_t(4096), C.byref(plen))
    assert rc == 0
    assert com.raw == want_com and proof.raw[: plen.value] == want_proof
    assert oracle.gadget_verify(kind, param, com.raw, proof.raw[: plen.value], hashlib.shake_256(b"r").digest(64))
    # the device prover's phase functions on the host (prover_dev.hpp)
    com2 = C.create_string_buffer(32 * m)
    proof2 = C.create_string_buffer(4096)
    plen2 = C.c_size_t(0)
    for staged in (0, 1):                                # (1: stage by stage, as the device runs a phase)
        host.zkhost_set_pv_staged(staged)
        try:
            rc = host.zkhost_prove_dev_r1cs(GADGET_LABEL, m, n1, n, len(labels), lab, len(cons), (C.c_uint64 * len(offs))(*offs),
                                            (C.c_uint8 * nt)(*kinds), (C.c_uint32 * nt)(*idx), coeff, (C.c_int32 * nt)(*chal),
                                            (C.c_uint32 * nt)(*power), (C.c_uint32 * max(len(mult_def), 1))(*mult_def),
                                            b"".join(v.to_bytes(32, "little") for v in values), gv, C.c_size_t(len(given)), seed, gens,
                                            C.c_size_t(cap), com2, proof2, C.c_size_t(4096), C.byref(plen2))
        finally:
            host.zkhost_set_pv_staged(0)
        assert rc == 0
        assert com2.raw == want_com and proof2.raw[: plen2.value] == want_proof, staged


@pytest.mark.parametrize("seed,shape", [(11, (1, 0, 0)), (12, (2, 1, 0)), (13, (3, 0, 2)), (14, (3, 6, 3)), (15, (2, 9, 2))])
def test_random_described_systems_device_prover_functions_equal_host_prover(host, oracle, seed, shape):
    """Random satisfiable constraint systems as data (tests/gpu_util.py: random_system): the device prover's phase functions
    run on the host (prover_dev.hpp) and the host prover (r1cs_prover.hpp: desc_prover) produce the same bytes."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from gpu_util import random_system
    rng = random.Random(seed)
    m, n1, n2 = shape
    (m, n1, n, labels, cons), mult_def, values, given = random_system(rng, m, n1, n2, 1 if n2 else 0)
    cap = 1
    while cap < max(n, 1):
        cap *= 2
    gens = _gens_bytes(oracle, cap)
    offs, kinds, idx, coeff, chal, power = [0], [], [], b"", [], []
    for con in cons:
        for (k_, i_, c_, ch_, pw_) in con:
            kinds.append(k_); idx.append(i_); coeff += (c_ % L).to_bytes(32, "little"); chal.append(ch_); power.append(pw_)
        offs.append(len(kinds))
    nt = max(len(kinds), 1)
    lab = (C.c_char_p * max(len(labels), 1))(*labels)
    gv = b"".join(a.to_bytes(32, "little") + b.to_bytes(32, "little") for a, b in given)
    vb = b"".join(v.to_bytes(32, "little") for v in values)
    seed_b = hashlib.sha256(b"random system %d" % seed).digest()
    outs = []
    for fn, staged in ((host.zkhost_r1cs_prove, 0), (host.zkhost_prove_dev_r1cs, 0), (host.zkhost_prove_dev_r1cs, 1)):
        com, proof, plen = C.create_string_buffer(32 * m), C.create_string_buffer(4096), C.c_size_t(0)
        host.zkhost_set_pv_staged(staged)
        try:
            rc = fn(b"random system", m, n1, n, len(labels), lab, len(cons), (C.c_uint64 * len(offs))(*offs), (C.c_uint8 * nt)(*kinds),
                    (C.c_uint32 * nt)(*idx), coeff, (C.c_int32 * nt)(*chal), (C.c_uint32 * nt)(*power),
                    (C.c_uint32 * max(len(mult_def), 1))(*mult_def), vb, gv, C.c_size_t(len(given)), seed_b, gens, C.c_size_t(cap), com, proof,
                    C.c_size_t(4096), C.byref(plen))
        finally:
            host.zkhost_set_pv_staged(0)
        assert rc == 0
        outs.append((com.raw, proof.raw[: plen.value]))
    assert outs[0] == outs[1] == outs[2] and len(outs[0][1]) > 400


def test_host_worker_pool(host):
    """host_pool.hpp (the persistent workers behind the VM stage of zkgpu_tx_verify_batch and the prover's witness rows):
    every index exactly once for sizes 0 .. 4097 and 2 .. 100 requested threads, also with four callers at once (one
    at a time gets the pool, the others wait), and across repeated calls on the sleeping workers."""
    host.zkhost_pool_selftest.restype = C.c_uint64
    host.zkhost_pool_selftest.argtypes = [C.c_uint32, C.c_uint32]
    assert host.zkhost_pool_selftest(3, 1) == 0
    assert host.zkhost_pool_selftest(3, 4) == 0
    # host_threads = 0: never more than the affinity mask allows, never more than the control group's CPU quota
    host.zkhost_usable_cpus.restype = C.c_int
    n = host.zkhost_usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            assert n <= max(1, round(int(quota) / int(period)))
    except (OSError, ValueError):
        pass
    assert host.zkhost_pool_selftest(1, 1) == 0


def test_ticket_bursts_are_cut_into_equal_device_batches(host):
    """csrc/ticket_cut.hpp, the policy of session.hpp's ticket_dispatch (round 6): what is queued of one
    shape leaves in round(queued / target) device batches, every ticket in exactly one, in queue order, sizes equal to within
    one ticket; uniform bursts of 16 / 20 / 24 / 37 at a target of ten tickets are 8 + 8, 10 + 10, 12 + 12, 10 + 9 + 9 + 9;
    a batch never exceeds 1.5 targets + one ticket; drawn ticket sizes obey the same invariants."""
    import random
    host.zkhost_ticket_cut.restype = C.c_size_t

    def cut(sizes, target):
        arr = (C.c_uint64 * len(sizes))(*sizes)
        out = (C.c_uint64 * (len(sizes) + 1))()
        n = host.zkhost_ticket_cut(arr, C.c_size_t(len(sizes)), C.c_uint64(target), out, C.c_size_t(len(sizes) + 1))
        return [int(out[i]) for i in range(n)]

    for burst, want in ((16, [8, 8]), (20, [10, 10]), (24, [12, 12]), (37, [10, 9, 9, 9]), (25, [9, 8, 8]), (4, [4]), (1, [1]), (14, [14]), (15, [8, 7])):
        assert cut([1024] * burst, 10240) == want, burst
    assert cut([], 10240) == []
    assert cut([11 * 1024], 10240) == [1]                                # a ticket larger than the target is a batch of its own
    rng = random.Random(66)
    for _ in range(300):
        target = rng.choice([480, 4096, 10240])
        sizes = [rng.choice([1, 7, 48, 64, 1024, 3000]) for _ in range(rng.randrange(1, 60))]
        got = cut(sizes, target)
        total = sum(sizes)
        assert sum(got) == len(sizes) and all(g >= 1 for g in got)
        parts = max(1, (total + target // 2) // target)
        assert len(got) <= parts                                             # (a big ticket may finish a later part's share early)
        at, batch_tx = 0, []
        for g in got:
            batch_tx.append(sum(sizes[at: at + g]))
            at += g
        assert max(batch_tx) <= (3 * target) // 2 + max(sizes) + target // max(parts, 1)
        if len(set(sizes)) == 1 and len(got) == parts:                       # uniform tickets: equal to within one ticket
            assert max(got) - min(got) <= 1
