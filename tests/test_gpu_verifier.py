"""GPU parity for the proof-bytes boundary: zkgpu_cloak_verify_batch (host transcript replay in C++
+ device MSMs) vs the oracle's Verifier on the same proof bytes; and the product's MSM terms vs the
oracle's."""
import hashlib

import pytest

from gpu_util import bits

pytestmark = pytest.mark.gpu
L = 2**252 + 27742317777372353535851937790883648493


@pytest.fixture(scope="module")
def ctx():
    from zkvm_amd import Context
    c = Context(0)
    yield c
    c.close()


def _txs(oracle, count, n_in, n_out, seed):
    from zkvm_amd.verifier import CloakTx
    com, proofs = oracle.cloak_prove_batch(count, n_in, n_out, seed, threads=8)
    w = 64 * (n_in + n_out)
    return [CloakTx(n_in, n_out, com[w * i: w * (i + 1)], proofs[i]) for i in range(count)]


@pytest.mark.parametrize("table_bits", [0, 9])
def test_cloak_verify_batch_vs_oracle(ctx, oracle, table_bits):
    from zkvm_amd.verifier import BulletproofGens, CloakTx, InvalidR1CSProof, Verifier
    gens = BulletproofGens(ctx, 256, table_bits=table_bits)
    txs = _txs(oracle, 24, 2, 2, b"\x01" * 32) + _txs(oracle, 6, 1, 2, b"\x02" * 32) + _txs(oracle, 6, 3, 3, b"\x03" * 32)
    # corruptions: proof byte, commitment byte, truncated proof, identity T_1, non-canonical t_x, wrong shape
    def mut(tx, **kw):
        return CloakTx(kw.get("n_in", tx.n_in), kw.get("n_out", tx.n_out), kw.get("commitments", tx.commitments),
                       kw.get("proof", tx.proof))
    p = bytearray(txs[3].proof); p[1 + 32 * 12 + 3] ^= 0x10; txs[3] = mut(txs[3], proof=bytes(p))
    cm = bytearray(txs[5].commitments); cm[70] ^= 1; txs[5] = mut(txs[5], commitments=bytes(cm))
    txs[7] = mut(txs[7], proof=txs[7].proof[:-32])
    p = bytearray(txs[9].proof); p[1 + 32 * 6: 1 + 32 * 7] = bytes(32); txs[9] = mut(txs[9], proof=bytes(p))
    p = bytearray(txs[11].proof); p[1 + 32 * 11: 1 + 32 * 12] = L.to_bytes(32, "little"); txs[11] = mut(txs[11], proof=bytes(p))
    txs[13] = mut(txs[13], proof=txs[14].proof)                       # someone else's proof
    p = bytearray(txs[15].proof); p[1 + 32 * 2: 1 + 32 * 3] = bytes.fromhex("01" + "00" * 31); txs[15] = mut(txs[15], proof=bytes(p))
    r = hashlib.shake_256(b"verifier r").digest(64 * len(txs))
    want = [int(oracle.cloak_verify(t.commitments, t.n_in, t.n_out, t.proof, r[64 * i: 64 * i + 64]))
            for i, t in enumerate(txs)]
    assert want.count(0) == 7 and want.count(1) == len(txs) - 7
    v = Verifier(ctx, gens, host_threads=4)
    bm = v.verify_bitmap(txs, r)
    assert bits(bm, len(txs)) == want
    res = v.verify_cloak_txs(txs, r)
    assert [x is None for x in res] == [bool(b) for b in want]
    assert all(isinstance(x, InvalidR1CSProof) for x in res if x is not None)
    # OS randomness for r: same verdicts
    assert bits(v.verify_bitmap(txs), len(txs)) == want
    assert v.verify_bitmap([]) == b""
    gens.close()


def test_generators_capacity_too_small_rejects(ctx, oracle):
    from zkvm_amd.verifier import BulletproofGens, Verifier
    gens = BulletproofGens(ctx, 128)           # 2-in/2-out needs 256
    txs = _txs(oracle, 2, 2, 2, b"\x04" * 32)
    assert Verifier(ctx, gens).verify_bitmap(txs) == b"\x00"   # InvalidGeneratorsLength in the reference
    gens.close()


@pytest.mark.parametrize("table_bits", [0, 9])
def test_device_side_preparation_equals_host_and_oracle(ctx, oracle, table_bits):
    """zkgpu_cloak_verify_batch_gpu (Merlin replay + scalar preparation on the device)
    gives the verdicts of the host-prepared path and of the oracle on the same proof bytes."""
    from zkvm_amd.verifier import BulletproofGens, CloakTx, Verifier
    gens = BulletproofGens(ctx, 256, table_bits=table_bits)
    txs = _txs(oracle, 70, 2, 2, b"\x11" * 32) + _txs(oracle, 5, 1, 1, b"\x12" * 32) + _txs(oracle, 5, 3, 3, b"\x13" * 32) \
        + _txs(oracle, 4, 1, 2, b"\x14" * 32)

    def mut(tx, **kw):
        return CloakTx(kw.get("n_in", tx.n_in), kw.get("n_out", tx.n_out), kw.get("commitments", tx.commitments),
                       kw.get("proof", tx.proof))
    p = bytearray(txs[3].proof); p[1 + 32 * 12 + 3] ^= 0x10; txs[3] = mut(txs[3], proof=bytes(p))          # t_x_blinding bit
    cm = bytearray(txs[5].commitments); cm[70] ^= 1; txs[5] = mut(txs[5], commitments=bytes(cm))
    txs[7] = mut(txs[7], proof=txs[7].proof[:-32])                                                          # wrong length
    p = bytearray(txs[9].proof); p[1 + 32 * 6: 1 + 32 * 7] = bytes(32); txs[9] = mut(txs[9], proof=bytes(p))  # T_1 identity
    p = bytearray(txs[11].proof); p[1 + 32 * 11: 1 + 32 * 12] = L.to_bytes(32, "little"); txs[11] = mut(txs[11], proof=bytes(p))
    txs[13] = mut(txs[13], proof=txs[14].proof)
    p = bytearray(txs[15].proof); p[0] = 0; txs[15] = mut(txs[15], proof=bytes(p))                          # version byte
    p = bytearray(txs[17].proof); p[1 + 32 * 14 + 32 * 5: 1 + 32 * 14 + 32 * 6] = bytes(32); txs[17] = mut(txs[17], proof=bytes(p))  # an IPA point = identity
    p = bytearray(txs[19].proof); p[-1] |= 0x80; txs[19] = mut(txs[19], proof=bytes(p))                     # b not canonical
    p = bytearray(txs[72].proof); p[40] ^= 4; txs[72] = mut(txs[72], proof=bytes(p))                        # in the 1x1 group
    r = hashlib.shake_256(b"verifier r 2").digest(64 * len(txs))
    want = [int(oracle.cloak_verify(t.commitments, t.n_in, t.n_out, t.proof, r[64 * i: 64 * i + 64]))
            for i, t in enumerate(txs)]
    assert want.count(0) == 10
    v = Verifier(ctx, gens, host_threads=4)
    assert bits(v.verify_bitmap(txs, r), len(txs)) == want
    assert bits(v.verify_bitmap_gpu(txs, r), len(txs)) == want
    assert bits(v.verify_bitmap_gpu(txs), len(txs)) == want          # OS randomness
    info = v.plan_info(2, 2)
    assert info["multipliers"] == 150 and info["padded_n"] == 256 and info["proof_len"] == 1025
    v.close()
    gens.close()


def test_batches_in_flight_equal_synchronous_calls(ctx, oracle):
    """zkgpu_ctx_fork + zkgpu_cloak_verify_submit_dev / zkgpu_verify_wait: three batches in flight on
    forked contexts (shared chip-filling streams) give, batch by batch, the verdicts of the oracle and
    of the synchronous call; a second submit on a busy context is refused; the MSM-boundary submit too."""
    from zkvm_amd import ZkGpuError
    from zkvm_amd.verifier import BulletproofGens, Verifier
    gens = BulletproofGens(ctx, 256, table_bits=9)
    v = Verifier(ctx, gens)
    forks = [ctx, ctx.fork(), ctx.fork()]
    batches = []
    for b, count in enumerate((40, 17, 64)):
        txs = _txs(oracle, count, 2, 2, bytes([0x21 + b]) * 32)
        proofs = [bytearray(t.proof) for t in txs]
        coms = [bytearray(t.commitments) for t in txs]
        proofs[b + 1][1 + 32 * 11 + 2] ^= 4            # t_x
        coms[b + 3][5] ^= 1                             # a commitment
        proofs[b + 5][0] = 2                            # wire-format version
        r = hashlib.shake_256(b"in flight %d" % b).digest(64 * count)
        want = [int(oracle.cloak_verify(bytes(coms[i]), 2, 2, bytes(proofs[i]), r[64 * i: 64 * i + 64])) for i in range(count)]
        assert want.count(0) == 3
        batches.append((count, ctx.to_device(b"".join(coms)), ctx.to_device(b"".join(proofs)), ctx.to_device(r), want,
                        len(proofs[0])))
    for rounds in range(3):
        for c, (count, d_com, d_pr, d_r, want, plen) in zip(forks, batches):
            v.submit_packed_gpu_dev(2, 2, count, d_com, d_pr, plen, d_r, ctx=c)
        with pytest.raises(ZkGpuError):
            v.submit_packed_gpu_dev(2, 2, batches[0][0], batches[0][1], batches[0][2], batches[0][5], batches[0][3], ctx=forks[0])
        for c, (count, d_com, d_pr, d_r, want, plen) in zip(forks, batches):
            assert bits(c.verify_wait(), count) == want
    for c, (count, d_com, d_pr, d_r, want, plen) in zip(forks, batches):
        assert bits(v.verify_packed_gpu_dev(2, 2, count, d_com, d_pr, plen, d_r), count) == want
    with pytest.raises(ZkGpuError):
        forks[1].verify_wait()                          # nothing pending
    for (count, d_com, d_pr, d_r, want, plen) in batches:
        for d in (d_com, d_pr, d_r):
            ctx.free_device(d)
    v.close()
    for c in forks[1:]:
        c.close()
    gens.close()


@pytest.mark.parametrize("group", [1, 3, 16, 64])
def test_group_checks_give_per_transaction_verdicts(ctx, oracle, group):
    """zkgpu_set_group_size: whatever the group size, the accept bitmap is the oracle's per-transaction
    one -- clean batch, a batch with bad transactions spread over several groups (bad proof scalar, bad
    commitment, undecodable proof point, non-canonical scalar, wrong version), a batch of only bad ones,
    and a batch size that is not a multiple of the group."""
    from zkvm_amd.verifier import BulletproofGens, CloakTx, Verifier
    gens = BulletproofGens(ctx, 256, table_bits=8)
    v = Verifier(ctx, gens)
    ctx.set_group_size(group)
    try:
        base = _txs(oracle, 77, 2, 2, b"\x31" * 32)
        plen = len(base[0].proof)

        def run(txs, tag):
            r = hashlib.shake_256(b"group " + tag).digest(64 * len(txs))
            want = [int(oracle.cloak_verify(t.commitments, 2, 2, t.proof, r[64 * i: 64 * i + 64])) for i, t in enumerate(txs)]
            got = bits(v.verify_packed_gpu(2, 2, len(txs), b"".join(t.commitments for t in txs),
                                           b"".join(t.proof for t in txs), plen, r), len(txs))
            assert got == want, tag
            return want

        assert run(base, b"clean") == [1] * 77
        bad = list(base)
        def mut(i, off, val=None, com=False):
            t = bad[i]
            buf = bytearray(t.commitments if com else t.proof)
            if val is None:
                buf[off] ^= 0x40
            else:
                buf[off: off + len(val)] = val
            bad[i] = CloakTx(2, 2, bytes(buf) if com else t.commitments, t.proof if com else bytes(buf))
        mut(0, 1 + 32 * 11 + 1)                                   # t_x
        mut(5, 9, com=True)                                       # a commitment
        mut(17, 1 + 32 * 3, bytes([0xFF] * 32))                   # A_I2: not a ristretto encoding
        mut(18, 1 + 32 * 12, L.to_bytes(32, "little"))            # t_x_blinding = l (non-canonical)
        mut(40, 0, bytes([7]))                                    # version byte
        mut(76, 1 + 32 * (14 + 3) + 2)                            # an L/R point of the inner-product argument
        want = run(bad, b"spread")
        assert want.count(0) >= 5
        assert run([bad[i] for i in (0, 5, 17, 18, 40)], b"all bad") == [0] * 5
        assert run(bad[:group + 1] if group > 1 else bad[:2], b"ragged")[0] == 0
    finally:
        ctx.set_group_size(16)
        v.close()
        gens.close()


def test_larger_shape_4x4_on_device(ctx, oracle):
    """4-in/4-out cloak: 312 multipliers, padded n = 512, k = 9, 12 second-phase challenges -- the shape
    whose plan needs the most LDS and flattening passes -- through the device-side verifier, with and
    without group checks, against the oracle (SURVEY.md sec 8(d) config 4 draws from these shapes)."""
    from zkvm_amd.verifier import BulletproofGens, CloakTx, Verifier
    gens = BulletproofGens(ctx, 512, table_bits=6)
    v = Verifier(ctx, gens)
    txs = _txs(oracle, 9, 4, 4, b"\x44" * 32)
    p = bytearray(txs[4].proof); p[1 + 32 * 13 + 7] ^= 2; txs[4] = CloakTx(4, 4, txs[4].commitments, bytes(p))   # e_blinding
    r = hashlib.shake_256(b"4x4").digest(64 * len(txs))
    want = [int(oracle.cloak_verify(t.commitments, 4, 4, t.proof, r[64 * i: 64 * i + 64])) for i, t in enumerate(txs)]
    assert want == [1, 1, 1, 1, 0, 1, 1, 1, 1]
    info = v.plan_info(4, 4)
    assert info["padded_n"] == 512 and info["proof_len"] == len(txs[0].proof)
    try:
        for group in (16, 1, 4):
            ctx.set_group_size(group)
            assert bits(v.verify_bitmap_gpu(txs, r), len(txs)) == want, group
    finally:
        ctx.set_group_size(16)
        v.close()
        gens.close()


def _witness(rng, n_in, n_out, two):
    fl = [rng.randrange(2**250).to_bytes(32, "little") for _ in range(2)]
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
    return q_in + q_out, f_in + f_out


@pytest.mark.parametrize("shape,table_bits", [((2, 2), 8), ((1, 1), 8), ((3, 2), 8), ((2, 2), 16)])
def test_gpu_prover_equals_oracle_prover_and_verifies(ctx, oracle, shape, table_bits):
    """zkgpu_cloak_prove_batch (config 5 / sec 8 row f-4): every commitment and proof byte equals the
    oracle prover's on the same witness and seed; the device-side verifier and the oracle accept them;
    an unbalanced witness yields a proof that both reject.  (16-bit tables: what bench.py proves over -- from 15 bits on
    k_static_digits folds the a_R = -1 of every bit multiplier into one digit on the negated generator.)"""
    import random
    from zkvm_amd.verifier import BulletproofGens, Prover, Verifier
    n_in, n_out = shape
    rng = random.Random(77 + 10 * n_in + n_out)
    gens = BulletproofGens(ctx, 256, table_bits=table_bits)
    batch = 21
    qs, fs, seeds = [], [], []
    for i in range(batch):
        q, f = _witness(rng, n_in, n_out, two=(i & 1) and n_in >= 2 and n_out >= 2)
        qs.append(q); fs.append(f); seeds.append(hashlib.sha256(b"gpu prover %d" % i).digest())
    qs[5] = list(qs[5]); qs[5][-1] += 1                        # unbalanced: outputs exceed inputs by one
    try:
        for mode in (1, 0, 16 + 4):                             # host threads in lockstep; the whole proof on the device; the same
            ctx.set_prover_mode(mode)                           # cut into four slices in flight (21 statements: 5 + 5 + 5 + 6)
            txs = Prover(ctx, gens, host_threads=8).prove(n_in, n_out, qs, fs, seeds)
            for i in range(batch):
                rc, want_com, want_proof, _ = oracle.cloak_prove(qs[i], fs[i], n_in, n_out, seeds[i])
                assert rc == 0 and txs[i].commitments == want_com and txs[i].proof == want_proof, (mode, i)
    finally:
        ctx.set_prover_mode(0)
    r = hashlib.shake_256(b"prover r").digest(64 * batch)
    want = [int(oracle.cloak_verify(t.commitments, n_in, n_out, t.proof, r[64 * i: 64 * i + 64])) for i, t in enumerate(txs)]
    assert want == [0 if i == 5 else 1 for i in range(batch)]
    v = Verifier(ctx, gens)
    assert bits(v.verify_bitmap_gpu(txs, r), batch) == want
    v.close()
    gens.close()


def test_full_size_batches_vs_oracle(ctx, oracle):
    """BASELINE.json configs[1] size (1024) and beyond (4096): the committed golden proofs replicated under
    per-transaction verifier randomness with a sprinkle of corruptions, complete verification on the device
    (group checks on), every accept bit against the oracle's full verifier (OpenMP)."""
    import os
    import struct
    from zkvm_amd.verifier import BulletproofGens, Verifier
    raw = open(os.path.join(os.path.dirname(__file__), "golden", "cloak_2x2_proofs.bin"), "rb").read()
    count, n_in, n_out, plen = struct.unpack("<IIII", raw[8:24])
    w = 64 * (n_in + n_out)
    fix = [(raw[24 + (w + plen) * i: 24 + (w + plen) * i + w], raw[24 + (w + plen) * i + w: 24 + (w + plen) * (i + 1)])
           for i in range(count)]
    gens = BulletproofGens(ctx, 256, table_bits=13)
    v = Verifier(ctx, gens)
    try:
        for batch in (1024, 4096):
            coms, proofs = [], []
            for i in range(batch):
                com, proof = fix[(i * 7 + batch) % count]
                if i % 97 == 5:
                    p = bytearray(proof); p[1 + 32 * (11 + i % 3) + (i % 31)] ^= 1 << (i % 8); proof = bytes(p)
                if i % 389 == 11:
                    cm = bytearray(com); cm[i % w] ^= 0x20; com = bytes(cm)
                coms.append(com); proofs.append(proof)
            r = hashlib.shake_256(b"full size %d" % batch).digest(64 * batch)
            want = list(oracle.cloak_verify_batch(b"".join(coms), n_in, n_out, b"".join(proofs), plen, r, threads=8))
            assert 0 < want.count(0) < batch // 20
            got = bits(v.verify_packed_gpu(n_in, n_out, batch, b"".join(coms), b"".join(proofs), plen, r), batch)
            assert got == want
    finally:
        v.close()
        gens.close()
