"""Oracle R1CS prover/verifier + cloak gadget: prove -> verify round trips, every mutation rejected.
Reference rows: SURVEY.md sec 8(a) a8, a9 (self-consistency only: this layer is parity-unpinned)."""
import hashlib

import pytest

L = 2**252 + 27742317777372353535851937790883648493


def fl(x):
    return (x % L).to_bytes(32, "little")


R = hashlib.shake_256(b"r-bytes").digest(64)


def test_two_in_two_out_shape_and_roundtrip(oracle):
    rc, com, proof, nm = oracle.cloak_prove([5, 9, 11, 3], [fl(7)] * 4, 2, 2, bytes(range(32)))
    assert rc == 0
    assert nm == 150                      # -> padded n = 256, k = 8, m = 8  (SURVEY.md sec 8(a) "n for 2-in/2-out")
    assert len(proof) == oracle.cloak_proof_size(256) == 1 + 32 * (14 + 16 + 2)
    assert oracle.cloak_verify(com, 2, 2, proof, R)
    assert oracle.cloak_verify(com, 2, 2, proof, bytes(64))       # any verifier weight r
    prep = oracle.cloak_verify_prepare(com, 2, 2, proof, R)
    assert len(prep[0]) == 32 * 35 and len(prep[2]) == 32 * 514 and prep[3] == 256


@pytest.mark.parametrize("n_in,n_out,q,flv", [
    (1, 1, [42, 42], [3, 3]),
    (1, 2, [10, 4, 6], [3, 3, 3]),
    (2, 1, [10, 4, 14], [3, 3, 3]),
    (2, 2, [5, 9, 5, 9], [7, 8, 7, 8]),
    (3, 3, [1, 2, 3, 3, 2, 1], [5, 5, 9, 9, 5, 5]),
    (2, 3, [2**63, 2**63 - 1, 2**64 - 1, 0, 0], [4, 4, 4, 4, 4]),
])
def test_shapes_accept(oracle, n_in, n_out, q, flv):
    rc, com, proof, _ = oracle.cloak_prove(q, [fl(x) for x in flv], n_in, n_out, bytes([n_in, n_out] * 16))
    assert rc == 0 and oracle.cloak_verify(com, n_in, n_out, proof, R)


@pytest.mark.parametrize("q,flv", [
    ([5, 9, 11, 4], [7, 7, 7, 7]),        # creates value
    ([5, 9, 9, 5], [7, 8, 7, 8]),         # moves value across flavors
    ([5, 9, 14, 0], [7, 7, 7, 9]),        # new flavor with zero quantity
])
def test_unbalanced_witness_rejected(oracle, q, flv):
    rc, com, proof, _ = oracle.cloak_prove(q, [fl(x) for x in flv], 2, 2, bytes(32))
    assert rc == 0 and not oracle.cloak_verify(com, 2, 2, proof, R)


def test_every_mutation_rejected(oracle):
    rc, com, proof, _ = oracle.cloak_prove([5, 9, 11, 3], [fl(7)] * 4, 2, 2, bytes(range(32)))
    for off in [0, 1, 33, 65, 1 + 32 * 6, 1 + 32 * 10, 1 + 32 * 11, 1 + 32 * 13, 1 + 32 * 14, 1 + 32 * 29, len(proof) - 40, len(proof) - 1]:
        bad = bytearray(proof)
        bad[off] ^= 2
        assert not oracle.cloak_verify(com, 2, 2, bytes(bad), R), off
    for off in [0, 31, 32, 64 * 3 + 5]:
        badc = bytearray(com)
        badc[off] ^= 1
        assert not oracle.cloak_verify(bytes(badc), 2, 2, proof, R), off
    assert not oracle.cloak_verify(com, 2, 2, proof[:-32], R)         # truncated
    assert not oracle.cloak_verify(com, 2, 2, proof + bytes(32), R)   # padded
    assert not oracle.cloak_verify(com, 2, 1, proof, R)               # wrong statement
    # identity where validate_and_append_point forbids it (T_1 := identity)
    bad = bytearray(proof)
    bad[1 + 32 * 6: 1 + 32 * 7] = bytes(32)
    assert oracle.cloak_verify_prepare(com, 2, 2, bytes(bad), R) is None
    # non-canonical scalar t_x (l itself)
    bad = bytearray(proof)
    bad[1 + 32 * 11: 1 + 32 * 12] = L.to_bytes(32, "little")
    assert oracle.cloak_verify_prepare(com, 2, 2, bytes(bad), R) is None


def test_batch_prover_is_deterministic_and_valid(oracle):
    com, proofs = oracle.cloak_prove_batch(6, 2, 2, b"\x07" * 32, threads=2)
    com2, proofs2 = oracle.cloak_prove_batch(6, 2, 2, b"\x07" * 32, threads=1)
    assert com == com2 and proofs == proofs2
    for i, p in enumerate(proofs):
        assert oracle.cloak_verify(com[256 * i: 256 * (i + 1)], 2, 2, p, R)
