"""Keccak / STROBE / Merlin / generator derivation: C oracle vs pyref vs hashlib and public constants.
Reference rows: SURVEY.md sec 8(a) a10, a11."""
import hashlib
import random

# merlin's published "test protocol" vector (also the first test of gtank/merlin).
MERLIN_KAT = "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
# dalek's published `PedersenGens::default().B_blinding` (compressed)
B_BLINDING = "8c9240b456a9e6dc65c377a1048d745f94a08cdb7f44cbcd7b46f34048871134"


def test_keccak_vs_hashlib(oracle, pyref):
    rng = random.Random(7)
    for n in [0, 1, 31, 71, 72, 73, 135, 136, 137, 199, 200, 1000]:
        d = bytes(rng.getrandbits(8) for _ in range(n))
        assert oracle.sha3_512(d) == hashlib.sha3_512(d).digest()
        assert oracle.shake256(d, 300) == hashlib.shake_256(d).digest(300)
        if n < 140:
            assert pyref.sha3_512(d) == hashlib.sha3_512(d).digest()
            assert pyref.shake256(d, 200) == hashlib.shake_256(d).digest(200)


def test_merlin_known_answer(oracle, pyref):
    t = oracle.MerlinTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == MERLIN_KAT
    t2 = pyref.Transcript(b"test protocol")
    t2.append_message(b"some label", b"some data")
    assert t2.challenge_bytes(b"challenge", 32).hex() == MERLIN_KAT


def test_merlin_c_vs_python_long_sequences(oracle, pyref):
    rng = random.Random(8)
    tc, tp = oracle.MerlinTranscript(b"ZkVM.r1cs-test"), pyref.Transcript(b"ZkVM.r1cs-test")
    for i in range(40):
        label = b"l%d" % i
        msg = bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 1, 32, 64, 165, 166, 167, 400])))
        tc.append_message(label, msg)
        tp.append_message(label, msg)
        if i % 3 == 0:
            n = rng.choice([1, 32, 64, 200])
            assert tc.challenge_bytes(b"c", n) == tp.challenge_bytes(b"c", n)
        if i % 7 == 0:
            tc.append_u64(b"n", i * 1000003)
            tp.append_u64(b"n", i * 1000003)
            assert tc.challenge_scalar(b"s") == tp.challenge_scalar(b"s")


def test_generators(oracle, pyref):
    b, bb = oracle.pedersen_gens()
    assert b.hex() == "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76"
    assert bb.hex() == B_BLINDING
    gp, hp = pyref.bulletproof_gens(6)
    assert oracle.bulletproof_gens(6, "G") == [pyref.encode(p) for p in gp]
    assert oracle.bulletproof_gens(6, "H") == [pyref.encode(p) for p in hp]
    # chains are prefixes of each other and differ per party / per side
    g8 = oracle.bulletproof_gens(8, "G")
    assert g8[:6] == oracle.bulletproof_gens(6, "G")
    assert g8 != oracle.bulletproof_gens(8, "H") and g8 != oracle.bulletproof_gens(8, "G", party=1)
