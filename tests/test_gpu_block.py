"""GPU parity for whole blocks of mixed shapes (BASELINE.json configs[3]), the exchange step behind the C ABI,
the device-side verifier head byte by byte, the benched configuration, and the 2^20 MSM against its committed
expected value (configs[2])."""
import hashlib
import json
import os

import pytest

from gpu_util import L, bits, load_cloak_fixture, load_mixed_fixture, mixed_block, msm_2p20_inputs, oracle_block_bits, random_system

pytestmark = pytest.mark.gpu
R260_INV = pow(pow(2, 260, L), -1, L)


@pytest.fixture(scope="module")
def ctx():
    from zkvm_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def gens512(ctx):
    """ONE table set for every shape up to 4x4 (padded n = 512): 16-bit windows, 51.7 GB."""
    from zkvm_amd.verifier import BulletproofGens
    g = BulletproofGens(ctx, 512, table_bits=16)
    yield g
    g.close()


def _cloak(txs):
    from zkvm_amd.verifier import CloakTx
    return [CloakTx(a, b, c, p) for a, b, c, p in txs]


def test_mixed_arity_shard_8192_vs_oracle(ctx, gens512, oracle):
    """An 8 192-transaction shard of SURVEY.md sec 8(d) config 4 (shapes 1x1, 1x2, 2x2, 3x3, 4x4 drawn with a
    fixed seed, > 1 % corrupted, every kind of corruption in every shape) through zkgpu_verifier_verify
    (host memory) and zkgpu_txblock + zkgpu_verifier_verify_block (resident in HBM): every accept bit equals the
    oracle's full verifier."""
    from zkvm_amd.verifier import BlockVerifier
    n = 8192
    txs = mixed_block(n, seed=4)
    r = hashlib.shake_256(b"config 4 shard").digest(64 * n)
    want = oracle_block_bits(oracle, txs, r, threads=16)
    bad = [i for i in range(n) if not want[i]]
    assert n // 100 < len(bad) < n // 20
    assert {(txs[i][0], txs[i][1]) for i in bad} == set(load_mixed_fixture())      # every shape has rejected ones
    bv = BlockVerifier(ctx, gens512)
    assert 3 <= bv.lanes() <= 6              # six asked for; lanes that would not run beside the others are not kept
    try:
        assert bits(bv.verify(_cloak(txs), r), n) == want
        blk = bv.block(_cloak(txs), r)
        assert blk.shapes() >= 6              # five shapes + the truncated proofs
        for _ in range(2):
            assert bits(bv.verify_block(blk), n) == want
        blk.close()
        assert bits(bv.verify(_cloak(txs)), n) == want          # verifier randomness from getrandom(2)
        assert bv.verify([]) == b""
    finally:
        bv.close()


def test_blocks_in_flight_equal_the_oracle(ctx, gens512, oracle):
    """zkgpu_verifier_block_start / _finish: three different blocks started before any is finished, on a verifier with
    fewer lanes than batches (starting has to wait for the oldest batch, whose verdicts must stay with its own run),
    finished out of order; every accept bit equals the oracle's, a run id cannot be
    finished twice."""
    from zkvm_amd import ZkGpuError
    from zkvm_amd.verifier import BlockVerifier
    blocks = []
    for k, n in enumerate((700, 900, 400)):
        txs = mixed_block(n, seed=20 + k)
        r = hashlib.shake_256(b"blocks in flight %d" % k).digest(64 * n)
        want = oracle_block_bits(oracle, txs, r, threads=16)
        assert 0 < sum(want) < n
        blocks.append((txs, r, want))
    for lanes, chunk in ((3, 64), (10, 0)):
        bv = BlockVerifier(ctx, gens512, batches_in_flight=lanes, chunk=chunk)
        try:
            resident = [bv.block(_cloak(txs), r) for txs, r, _ in blocks]
            for _ in range(2):
                runs = [bv.block_start(b) for b in resident]
                assert len(set(runs)) == 3
                for k in (1, 0, 2):
                    assert bits(bv.block_finish(runs[k]), len(blocks[k][0])) == blocks[k][2]
                with pytest.raises(ZkGpuError):
                    bv.block_finish(runs[1])
                import ctypes
                scratch = ctypes.create_string_buffer(1024)
                assert bv.lib.zkgpu_verifier_block_finish(bv.h, runs[1], scratch) == -1       # ZKGPU_EINVAL, at the C ABI too
            # the same block twice in flight, and the plain call while a run is open
            a, b = bv.block_start(resident[0]), bv.block_start(resident[0])
            assert bits(bv.verify_block(resident[2]), len(blocks[2][0])) == blocks[2][2]
            assert bits(bv.block_finish(b), len(blocks[0][0])) == blocks[0][2]
            assert bits(bv.block_finish(a), len(blocks[0][0])) == blocks[0][2]
            for blk in resident:
                blk.close()
        finally:
            bv.close()


def test_one_phase_wire_format_on_the_device(ctx, gens512, oracle):
    """Proofs in the one-phase wire format (version byte 0, A_I2 A_O2 S2 left out: what upstream writes for a statement
    without a second phase, here the 1x1 cloak) through every device path -- a uniform batch, and a block mixing both
    forms and shapes -- against the oracle, which reads the short form as the long one with identities."""
    from zkvm_amd.verifier import Verifier, BlockVerifier
    fix = load_mixed_fixture()
    one = fix[(1, 1)][:48]
    short = []
    for i, (com, proof) in enumerate(one):
        assert proof[0] == 1 and proof[97:193] == bytes(96)
        sp = bytearray(b"\x00" + proof[1:97] + proof[193:])
        if i % 7 == 3:
            sp[1 + 32 * 8 + 5] ^= 2                               # t_x: only the multiscalar multiplication can tell
        if i % 11 == 5:
            sp[0] = 1                                            # version byte disagrees with the length
        short.append((com, bytes(sp)))
    n = len(short)
    r = hashlib.shake_256(b"one phase device").digest(64 * n)
    plen = len(short[0][1])
    want = list(oracle.cloak_verify_batch(b"".join(c for c, _ in short), 1, 1, b"".join(p for _, p in short), plen, r, threads=8))
    assert 0 < sum(want) < n
    v = Verifier(ctx, gens512)
    try:
        for group in (16, 1):
            ctx.set_group_size(group)
            assert bits(v.verify_packed_gpu(1, 1, n, b"".join(c for c, _ in short), b"".join(p for _, p in short), plen, r), n) == want
    finally:
        ctx.set_group_size(16)
        v.close()
    # a block: long and short 1x1 proofs, 2x2 proofs, and a 2x2 proof cut down to the short form (parsed, then rejected)
    two = fix[(2, 2)][:20]
    txs = [(1, 1, c, p) for c, p in one[:10]] + [(1, 1, c, p) for c, p in short[:30]] + [(2, 2, c, p) for c, p in two]
    txs.append((2, 2, two[0][0], b"\x00" + two[0][1][1:97] + two[0][1][193:]))
    rb = hashlib.shake_256(b"one phase block").digest(64 * len(txs))
    want_b = oracle_block_bits(oracle, txs, rb)
    assert want_b[-1] == 0 and sum(want_b) > 40
    bv = BlockVerifier(ctx, gens512)
    try:
        assert bits(bv.verify(_cloak(txs), rb), len(txs)) == want_b
    finally:
        bv.close()


def test_block_rejects_unverifiable_shapes_one_by_one(ctx, oracle):
    """Too few generators for a shape, a statement with no values, a proof of the wrong length: that transaction is
    rejected (InvalidGeneratorsLength / malformed proof in the reference), its neighbours are verified, and the
    verifier keeps working afterwards (nothing left pending on its contexts)."""
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens, CloakTx, Verifier
    fix = load_mixed_fixture()
    gens = BulletproofGens(ctx, 128, table_bits=8)        # 1x1 needs 64, 2x2 needs 256
    bv = BlockVerifier(ctx, gens, batches_in_flight=3)
    try:
        one = [CloakTx(1, 1, c, p) for c, p in fix[(1, 1)][:5]]
        two = [CloakTx(2, 2, c, p) for c, p in fix[(2, 2)][:3]]
        txs = [one[0], two[0], one[1], CloakTx(0, 0, b"", one[2].proof), two[1], CloakTx(1, 1, one[3].commitments, one[3].proof[:-32]),
               one[4], two[2]]
        r = hashlib.shake_256(b"bad shapes").digest(64 * len(txs))
        want = [1, 0, 1, 0, 0, 0, 1, 0]
        assert [int(oracle.cloak_verify(t.commitments, t.n_in, t.n_out, t.proof, r[64 * i: 64 * i + 64]))
                for i, t in enumerate(txs) if i in (0, 2, 6)] == [1, 1, 1]
        for _ in range(3):
            assert bits(bv.verify(txs, r), len(txs)) == want
        assert bits(Verifier(ctx, gens).verify_bitmap(txs, r), len(txs)) == want      # the host-prepared path agrees
    finally:
        bv.close()
        gens.close()


def test_exchange_step_behind_the_abi(ctx, oracle):
    """zkgpu_comm over RCCL in a world of one (the 1-GPU box): unique id, communicator, ncclAllGather of raw bytes
    and of bitmaps, the fail-closed status word, zkgpu_verifier_verify_sharded; and the RCCL-free world of one."""
    from zkvm_amd import ZkGpuError
    from zkvm_amd.native import Comm, shard_cuts
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    for uid in (Comm.unique_id(), None):
        comm = Comm(ctx, 0, 1, uid)
        payload = bytes(range(256)) * 5
        assert comm.allgather(payload) == payload
        n = 1000
        bm = hashlib.shake_256(b"bitmap").digest((n + 7) // 8 - 1) + b"\x7f"
        assert comm.allgather_bitmap([0, n], bm) == bm
        with pytest.raises(ZkGpuError) as e:
            comm.allgather_bitmap([0, n], bm, local_status=-3)
        assert e.value.code == -3
        comm.close()
    gens = BulletproofGens(ctx, 256, table_bits=8)
    bv = BlockVerifier(ctx, gens, batches_in_flight=2)
    comm = Comm(ctx, 0, 1, Comm.unique_id())
    try:
        txs = mixed_block(150, seed=9, bad_every=17)
        txs = [t for t in txs if (t[0], t[1]) != (4, 4)]          # 256 generators: no 4x4
        r = hashlib.shake_256(b"sharded").digest(64 * len(txs))
        want = oracle_block_bits(oracle, txs, r)
        assert bits(bv.verify_sharded(comm, _cloak(txs), r), len(txs)) == want
        assert shard_cuts([(t[0], t[1]) for t in txs], 1) == [0, len(txs)]
    finally:
        comm.close()
        bv.close()
        gens.close()


def _slot(cuts, rank, bitmap, status=0):
    """one rank's contribution to the gather, framed as csrc/comm_frame.hpp frames it (written down independently here)"""
    width = max((cuts[i + 1] - cuts[i] + 7) // 8 for i in range(len(cuts) - 1))
    slot = 8 + (width + 7) // 8 * 8
    n = cuts[rank + 1] - cuts[rank]
    body = bitmap[: (n + 7) // 8] if status == 0 else b""
    return (status & 0xFFFFFFFF).to_bytes(4, "little") + bytes(4) + body + bytes(slot - 8 - len(body))


@pytest.mark.parametrize("world,rank", [(2, 1), (3, 0), (8, 5)])
def test_exchange_step_in_a_mocked_world_of_n_ranks(ctx, oracle, world, rank):
    """zkgpu_comm_* at world 2 / 3 / 8 on ONE GPU: the collective function table replaced by an in-process mock whose other
    ranks contribute what the test supplies (zkgpu_debug_comm_mock), everything else -- buffers, stream, copies, framing,
    statuses, zkgpu_verifier_verify_sharded with its cuts -- the product's own path.  Every rank's bits land at their
    cuts; a peer's error, or the poison word of a peer whose copy failed, gives an error and an all-zero bitmap here; this
    rank's own fault is reported as its own code; and the sharded verification of a mixed block verifies exactly
    [cuts[rank], cuts[rank + 1]) on this GPU, the oracle's verdicts for the other shards arriving through the gather."""
    import ctypes as C
    from zkvm_amd import ZkGpuError
    from zkvm_amd.native import Comm, shard_cuts
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    lib = ctx.lib
    txs = [t for t in mixed_block(40 * world, seed=20 + world, bad_every=7) if (t[0], t[1]) != (4, 4)]
    n = len(txs)
    r = hashlib.shake_256(b"mock world %d" % world).digest(64 * n)
    want = oracle_block_bits(oracle, txs, r)
    cuts = shard_cuts([(t[0], t[1]) for t in txs], world)
    assert cuts[0] == 0 and cuts[-1] == n and all(a <= b for a, b in zip(cuts, cuts[1:]))

    def shard_bitmap(k):
        b = bytearray((cuts[k + 1] - cuts[k] + 7) // 8)
        for j, i in enumerate(range(cuts[k], cuts[k + 1])):
            b[j // 8] |= want[i] << (j % 8)
        return bytes(b)

    def install(peers):
        blob = b"".join(peers)
        assert lib.zkgpu_debug_comm_mock(ctx.h, world, blob, len(peers[0])) >= 0

    whole = bytearray((n + 7) // 8)
    for i, v in enumerate(want):
        whole[i // 8] |= v << (i % 8)
    gens = BulletproofGens(ctx, 256, table_bits=8)
    bv = BlockVerifier(ctx, gens, batches_in_flight=2)
    comm = None
    try:
        good = [_slot(cuts, k, shard_bitmap(k)) for k in range(world)]
        junk = list(good)
        junk[rank] = b"\xff" * len(good[0])                   # the mock overwrites this rank's place with what it really sends
        install(junk)
        comm = Comm(ctx, rank, world, bytes(128))
        before = lib.zkgpu_debug_comm_mock(ctx.h, world, b"".join(junk), len(junk[0]))
        assert comm.allgather_bitmap(cuts, shard_bitmap(rank)) == bytes(whole)
        assert bits(bv.verify_sharded(comm, _cloak(txs), r), n) == want
        assert lib.zkgpu_debug_comm_mock(ctx.h, world, b"".join(junk), len(junk[0])) == before + 2
        # this rank's own fault: its code, zeros; and the collective was still entered (the peers are not left waiting)
        with pytest.raises(ZkGpuError) as e:
            comm.allgather_bitmap(cuts, shard_bitmap(rank), local_status=-4)
        assert e.value.code == -4
        assert lib.zkgpu_debug_comm_mock(ctx.h, world, b"".join(junk), len(junk[0])) == before + 3
        peer = (rank + 1) % world
        for status in (-3, 0x80000001 - (1 << 32)):          # a peer's error code; a peer's poison word
            bad = list(junk)
            bad[peer] = _slot(cuts, peer, b"", status)
            install(bad)
            out = C.create_string_buffer(len(whole))
            rc = lib.zkgpu_comm_allgather_bitmap(comm.h, (C.c_uint64 * (world + 1))(*cuts), shard_bitmap(rank), 0, out)
            assert rc == -7 and out.raw == bytes(len(whole))
            with pytest.raises(ZkGpuError):
                bv.verify_sharded(comm, _cloak(txs), r)
        # cuts that go backwards are refused before the collective (the same answer on every rank)
        back = list(cuts)
        if world >= 2:
            back[1] = back[2] + 1 if world > 2 else n + 1
            rc = lib.zkgpu_comm_allgather_bitmap(comm.h, (C.c_uint64 * (world + 1))(*back), shard_bitmap(rank), 0, C.create_string_buffer(len(whole) + 8))
            assert rc == -1
    finally:
        if comm is not None:
            comm.close()
        lib.zkgpu_debug_comm_mock(ctx.h, 0, None, 0)
        bv.close()
        gens.close()


def test_batches_of_blocks_in_flight_are_merged_and_workspaces_can_be_reserved(ctx, gens512, oracle):
    """Since round 4 a block's batches are requests of the verifier's ticket queue: zkgpu_verifier_block_start only QUEUES them
    (they leave when the merge target is reached or a run is finished), and batches of one shape from DIFFERENT blocks in
    flight go to the device as one batch.  Four different mixed blocks started, finished in another order, twice (cold and
    warm), with zkgpu_verifier_reserve having sized every lane for the merged batches beforehand; tickets of the caller's
    own queued in between keep their own verdicts; every bit against the oracle."""
    from zkvm_amd import ZkGpuError
    from zkvm_amd.verifier import BlockVerifier
    blocks = []
    for k, n in enumerate((900, 1100, 700, 1000)):
        txs = mixed_block(n, seed=60 + k, bad_every=9)
        r = hashlib.shake_256(b"merged blocks %d" % k).digest(64 * n)
        want = oracle_block_bits(oracle, txs, r, threads=16)
        assert 0 < sum(want) < n
        blocks.append((txs, r, want))
    fix, n_in, n_out, plen = load_cloak_fixture()
    t_com, t_proofs = b"".join(fix[i][0] for i in range(64)), b"".join(fix[i][1] for i in range(64))
    t_r = hashlib.shake_256(b"ticket beside blocks").digest(64 * 64)
    bv = BlockVerifier(ctx, gens512, batches_in_flight=4)
    bv.set_merge(2048)
    try:
        for shape in ((1, 1), (1, 2), (2, 2), (3, 3), (4, 4)):
            bv.reserve(shape[0], shape[1], 2048)
        bv.reserve(64, 64, 10)                                       # a shape the generator set cannot serve: nothing to do
        with pytest.raises(ZkGpuError):
            bv.reserve(2, 2, 0)
        resident = [bv.block(_cloak(txs), r) for txs, r, _ in blocks]
        for order in ((2, 0, 3, 1), (0, 1, 2, 3)):
            runs = [bv.block_start(b) for b in resident]
            tk = bv.submit(n_in, n_out, 64, t_com, t_proofs, plen, t_r)
            for k in order:
                assert bits(bv.block_finish(runs[k]), len(blocks[k][0])) == blocks[k][2], k
            assert bits(bv.wait(tk), 64) == [1] * 64
        for b in resident:
            b.close()
    finally:
        bv.close()


def test_config4_at_full_size_in_a_mocked_world_of_eight(ctx, gens512, oracle):
    """BASELINE configs[3] in its stated size on the one GPU there is: 65 536 mixed-arity transactions (bench.py's own
    construction: gpu_util.mixed_block, 1 in 61 damaged, every kind in every shape), cut into EIGHT shards by
    zkgpu_shard_cuts, and zkgpu_verifier_verify_sharded run as every rank 0 .. 7 in turn -- each time this GPU verifies that
    rank's shard and the other seven bitmaps arrive through the exchange from the collective mock -- so that every one of
    the 65 536 transactions has been verified on the device through the product's sharded path (cuts, shard verification,
    framing, unpacking), and the whole bitmap on every rank equals the expectation.  The expectation is bench.py's
    (damaged -> 0, everything else 1) and is held against the oracle's full verifier for the first shard and a sample of
    every other.  What this cannot cover is RCCL's own rendezvous between eight processes."""
    from zkvm_amd.native import Comm, shard_cuts
    from zkvm_amd.verifier import BlockVerifier
    world, n = 8, 65536
    txs = mixed_block(n, seed=0x5A6B564D)                    # (bench.py --config 4 --gpus 8 builds exactly this block)
    want = [0 if i % 61 == 3 else 1 for i in range(n)]
    r = hashlib.shake_256(b"config 4 full size").digest(64 * n)
    cuts = shard_cuts([(t[0], t[1]) for t in txs], world)
    assert cuts[0] == 0 and cuts[-1] == n and all(4000 < b - a < 13000 for a, b in zip(cuts, cuts[1:]))
    # the constructed expectation against the oracle: the whole first shard, and 300 transactions of every other
    idx = list(range(cuts[0], cuts[1])) + [i for k in range(1, world) for i in range(cuts[k], cuts[k + 1], max(1, (cuts[k + 1] - cuts[k]) // 300))]
    idx += [i for i in range(3, n, 61 * 7)]                                         # and damaged ones everywhere
    idx = sorted(set(idx))
    got = oracle_block_bits(oracle, [txs[i] for i in idx], b"".join(r[64 * i: 64 * i + 64] for i in idx), threads=16)
    assert got == [want[i] for i in idx]

    def shard_bitmap(k):
        b = bytearray((cuts[k + 1] - cuts[k] + 7) // 8)
        for j, i in enumerate(range(cuts[k], cuts[k + 1])):
            b[j // 8] |= want[i] << (j % 8)
        return bytes(b)
    whole = bytearray((n + 7) // 8)
    for i, v in enumerate(want):
        whole[i // 8] |= v << (i % 8)
    good = [_slot(cuts, k, shard_bitmap(k)) for k in range(world)]
    lib = ctx.lib
    block = _cloak(txs)
    bv = BlockVerifier(ctx, gens512)
    try:
        for rank in range(world):
            junk = list(good)
            junk[rank] = b"\xff" * len(good[0])                    # (overwritten by what this rank really sends)
            assert lib.zkgpu_debug_comm_mock(ctx.h, world, b"".join(junk), len(junk[0])) >= 0
            comm = Comm(ctx, rank, world, bytes(128))
            try:
                assert bv.verify_sharded(comm, block, r) == bytes(whole), rank
            finally:
                comm.close()
    finally:
        lib.zkgpu_debug_comm_mock(ctx.h, 0, None, 0)
        bv.close()


def test_synchronous_calls_never_run_over_a_batch_in_flight(ctx, oracle):
    """A context with a submitted batch refuses every synchronous entry point (they share its status words and pinned
    result buffer), and the batch's verdicts are untouched; zkgpu_tx_verify_batch -- whose key and signature stages are
    synchronous calls on the verifier's root context = lane 0 -- first collects what the lanes have in flight: a block
    of all-bad proofs started before it must still finish as all zeros, tickets keep their own bits."""
    from gpu_util import load_tx_fixture
    from zkvm_amd import ZkGpuError
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens, Verifier
    fix, n_in, n_out, plen = load_cloak_fixture()
    gens = BulletproofGens(ctx, 256, table_bits=10)
    v = Verifier(ctx, gens)
    n = 300
    com = b"".join(fix[i][0] for i in range(n))
    proofs = bytearray(b"".join(fix[i][1] for i in range(n)))
    for i in range(0, n, 7):
        proofs[i * plen + 1 + 32 * 11 + 5] ^= 1
    r = hashlib.shake_256(b"pending guard").digest(64 * n)
    want = list(oracle.cloak_verify_batch(com, n_in, n_out, bytes(proofs), plen, r, threads=8))
    d = [ctx.to_device(x) for x in (com, bytes(proofs), r)]
    bv = None
    try:
        v.submit_packed_gpu_dev(n_in, n_out, n, d[0], d[1], plen, d[2])
        sc = (5).to_bytes(32, "little")
        pt = oracle.encode(oracle.basepoint())
        for call in (lambda: ctx.msm(sc, pt), lambda: ctx.verify_batch(sc, pt, [0, 1]), lambda: ctx.hash_to_points(bytes(64)),
                     lambda: ctx.decode_check(pt)):
            with pytest.raises(ZkGpuError) as e:
                call()
            assert e.value.code == -1
        assert bits(ctx.verify_wait(), n) == want
        assert ctx.msm(sc, pt) == oracle.encode(oracle.scalarmult(5, oracle.basepoint()))       # and works again afterwards
        # the transaction path beside a block and tickets in flight on the same verifier
        bv = BlockVerifier(ctx, gens, batches_in_flight=3)
        bv.set_tx_format(bv.TXFORMAT_RECOLLECTED_V1)
        txs = load_tx_fixture()[:200]
        all_bad = []
        for i in range(256):
            p = bytearray(fix[i][1]); p[1 + 32 * 12 + (i % 30)] ^= 1 << (i % 8)
            all_bad.append((n_in, n_out, fix[i][0], bytes(p)))
        rb = hashlib.shake_256(b"all bad").digest(64 * 256)
        assert oracle_block_bits(oracle, all_bad[:16], rb[: 64 * 16]) == [0] * 16
        blk = bv.block(_cloak(all_bad), rb)
        for _ in range(3):
            run = bv.block_start(blk)
            ticket = bv.submit_dev(n_in, n_out, n, d[0], d[1], plen, d[2])
            bm, st = bv.verify_txs(txs)
            assert bm == bytes([0xFF]) * 25 and not any(st)
            assert bv.block_finish(run) == bytes(32)
            assert bits(bv.wait(ticket), n) == want
        blk.close()
    finally:
        if bv is not None:
            bv.close()
        v.close()
        for x in d:
            ctx.free_device(x)
        gens.close()


def test_lanes_are_probed_for_hardware_queues_and_a_late_environment_is_noticed(ctx, oracle):
    """zkgpu_verifier_create keeps only lanes whose streams really run side by side.  In this process (GPU_MAX_HW_QUEUES
    set before HIP started: conftest.py did, on zkgpu_runtime_hint's advice) the context's stream pair overlaps and a ten-lane verifier keeps at least
    four lanes (the device runs fewer queues side by side than the runtime hands out: the rest are dropped, and counted);
    in a child process whose HIP runtime starts BEFORE the variable is set -- an embedding application that touched HIP
    first: the runtime's default of four queues, on which batches in flight take turns -- the library notices (the
    driver's device node is already open when zkgpu_init runs), says so, and verification still returns the right
    verdicts."""
    import subprocess
    import sys
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    assert ctx.queue_info()[0] == 1 and ctx.queue_info()[2] == 0
    gens = BulletproofGens(ctx, 256, table_bits=8)
    bv = BlockVerifier(ctx, gens, batches_in_flight=10)
    try:
        used_here, asked, dropped, late = bv.queue_info()
        assert asked == 10 and used_here + dropped == 10 and used_here >= 4 and late == 0, (used_here, asked, dropped, late)
        assert bv.lanes() == used_here
    finally:
        bv.close()
        gens.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import ctypes, os, sys, hashlib
os.environ.pop("GPU_MAX_HW_QUEUES", None)
hip = ctypes.CDLL("libamdhip64.so")
n = ctypes.c_int(0)
assert hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value >= 1
p = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(p), 4096) == 0          # the runtime is up now, with its default number of queues
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from gpu_util import bits, load_cloak_fixture
from zkvm_amd import Context
from zkvm_amd.verifier import BlockVerifier, BulletproofGens, CloakTx
ctx = Context(0)                                          # asks zkgpu_runtime_hint: too late, says the library
gens = BulletproofGens(ctx, 256, table_bits=8)
bv = BlockVerifier(ctx, gens, batches_in_flight=10)
used, asked, dropped, late = bv.queue_info()
assert ctx.queue_info()[2] == late
msg = bv.lib.zkgpu_verifier_last_error(bv.h).decode()
fix, n_in, n_out, plen = load_cloak_fixture()
txs = []
for i in range(300):
    com, proof = fix[i]
    if i %% 41 == 3:
        q = bytearray(proof); q[1 + 32 * 11 + 2] ^= 1; proof = bytes(q)
    txs.append(CloakTx(n_in, n_out, com, proof))
bv.lib.zkgpu_verifier_set_chunk(bv.h, 64)                  # many small batches: every lane gets work
got = bits(bv.verify(txs, hashlib.shake_256(b"late env").digest(64 * 300)), 300)
print("RESULT", used, asked, late, int(got == [0 if i %% 41 == 3 else 1 for i in range(300)]), os.environ.get("GPU_MAX_HW_QUEUES"), "|", msg)
""" % (root, root)
    out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line, (out.stdout[-2000:], out.stderr[-2000:])
    used, asked, late, correct = [int(x) for x in line[0].split()[1:5]]
    assert asked == 10 and correct == 1 and used >= 1
    assert late == 1 and "before GPU_MAX_HW_QUEUES was set" in line[0], line[0]


def test_a_second_verifier_on_the_device_is_created_with_a_warning_when_queues_are_many():
    """"nothing stops an integrator from creating two".  zkgpu_verifier_create now says so: status
    ZKGPU_WSECOND_VERIFIER (1, the verifier IS created and works) when another verifier is alive on the device and the
    runtime hands out 20 or more hardware queues -- the range in which two verifiers were measured to stall each other
     -- and plain 0 with 16 queues, or once the other one has been destroyed.  Fresh processes: the queue
    count is read when the HIP runtime starts."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = r"""
import os, sys, hashlib
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
from gpu_util import bits, load_cloak_fixture
from zkvm_amd import Context
from zkvm_amd.verifier import BlockVerifier, BulletproofGens, CloakTx
ctx = Context(0)
gens = BulletproofGens(ctx, 256, table_bits=8)
a = BlockVerifier(ctx, gens, batches_in_flight=3)
b = BlockVerifier(ctx, gens, batches_in_flight=3)
fix, n_in, n_out, plen = load_cloak_fixture()
txs = [CloakTx(n_in, n_out, *fix[i]) for i in range(40)]
ok = bits(b.verify(txs, hashlib.shake_256(b"second").digest(64 * 40)), 40) == [1] * 40
a.close()
c = BlockVerifier(ctx, gens, batches_in_flight=3)          # b is still alive
b.close(); c.close()
d = BlockVerifier(ctx, gens, batches_in_flight=3)          # nobody else is
print("RESULT", int(bool(a.warning)), int(bool(b.warning)), int(bool(c.warning)), int(bool(d.warning)), int(ok), "|", b.warning)
d.close(); gens.close(); ctx.close()
""" % (root, root)
    seen = {}
    for queues in ("24", "16"):
        env = dict(os.environ, GPU_MAX_HW_QUEUES=queues)
        out = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, timeout=600, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
        assert line, (out.stdout[-2000:], out.stderr[-2000:])
        seen[queues] = ([int(x) for x in line[0].split()[1:6]], line[0])
    assert seen["24"][0] == [0, 1, 1, 0, 1], seen["24"]
    assert "one verifier per process and device" in seen["24"][1]
    assert seen["16"][0] == [0, 0, 0, 0, 1], seen["16"]


def test_cooperative_keccak_primitives_and_permutation(ctx, oracle):
    """keccak_coop.hpp on the hardware: every cross-lane primitive (DPP row_ror:8 / row_shr:1 / row_shl:1,
    v_permlane16_swap, v_permlane32_swap, ds_bpermute) behaves as the host emulation assumes, and Keccak-f[1600] with
    one state per wavefront equals the oracle's on random states."""
    import ctypes as C
    import random
    rng = random.Random(5)
    a = [rng.getrandbits(32) for _ in range(64)]
    b = [rng.getrandbits(32) for _ in range(64)]
    addr = [4 * rng.randrange(64) for _ in range(64)]
    states = [[0] * 25] + [[rng.getrandbits(64) for _ in range(25)] for _ in range(200)]
    out, got = ctx.coop_selftest(a, b, addr, states)
    assert out[0] == [a[(i & ~15) | ((i + 8) & 15)] for i in range(64)]                      # row_ror:8
    assert out[1] == [a[i - 1] if i & 15 else a[i] for i in range(64)]                       # row_shr:1 (row-lane 0 keeps its value)
    assert out[2] == [a[i + 1] if (i & 15) != 15 else a[i] for i in range(64)]               # row_shl:1
    x, y = list(a), list(b)
    for row in (1, 3):
        for i in range(16):
            x[16 * row + i], y[16 * (row - 1) + i] = y[16 * (row - 1) + i], x[16 * row + i]
    assert out[3] == x and out[4] == y                                                       # v_permlane16_swap
    x, y = list(a), list(b)
    for i in range(32):
        x[32 + i], y[i] = y[i], x[32 + i]
    assert out[5] == x and out[6] == y                                                       # v_permlane32_swap
    assert out[7] == [a[addr[i] // 4] for i in range(64)]                                    # ds_bpermute
    lib = oracle.load()
    for st, g in zip(states, got):
        w = (C.c_uint64 * 25)(*st)
        lib.keccak_f1600(w)
        assert list(w) == g


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("group", [1, 16])
def test_device_verifier_head_bytes_equal_oracle(ctx, gens512, oracle, group, mode):
    """zkgpu_debug_read after a batch: the challenges of the device-side transcript replay (k_transcript) equal
    the oracle transcript's byte for byte, and every scalar of the verification equation the device prepares
    (k_prepare) equals the oracle's times the documented factor c' = rho y^(pn-1) prod u_j^2 -- for 1x1, 1x2, 2x2,
    3x3 and 4x4, transactions checked alone (rho = 1) and in groups (rho = r^2), with the transcript replayed one
    lane per transaction (k_transcript) and one wavefront per transaction (k_tape_gather, k_transcript_coop,
    k_challenges)."""
    from zkvm_amd.verifier import Verifier
    fix = load_mixed_fixture()
    v = Verifier(ctx, gens512)
    ctx.set_group_size(group)
    ctx.set_transcript_mode(mode)
    try:
        for (n_in, n_out), recs in sorted(fix.items()):
            batch = 5
            lay = v.plan_layout(n_in, n_out)
            info = v.plan_info(n_in, n_out)
            plen, pn, k, n2 = info["proof_len"], info["padded_n"], lay["k"], lay["n_chal2"]
            r = hashlib.shake_256(b"head %d %d" % (n_in, n_out)).digest(64 * batch)
            com = b"".join(c for c, _ in recs[:batch])
            proofs = b"".join(p for _, p in recs[:batch])
            assert bits(v.verify_packed_gpu(n_in, n_out, batch, com, proofs, plen, r), batch) == [1] * batch
            ch = ctx.debug_read("challenges", batch * lay["slots"] * 32)
            st = ctx.debug_read("static_scalars", batch * lay["n_static"] * 32)
            dy = ctx.debug_read("dyn_scalars", batch * lay["n_dyn"] * 32)
            for i in range(batch):
                c_i, p_i, r_i = recs[i][0], recs[i][1], r[64 * i: 64 * i + 64]
                want_ch = oracle.cloak_verify_challenges(c_i, n_in, n_out, p_i, r_i)
                assert len(want_ch) == n2 + 5 + k

                def slot(j):
                    o = (i * lay["slots"] + j) * 32
                    return int.from_bytes(ch[o: o + 32], "little") * R260_INV % L
                got_ch = [slot(14 + j) for j in range(n2)] + [slot(j) for j in range(5)] + [slot(14 + n2 + j) for j in range(k)]
                assert got_ch == want_ch, (n_in, n_out, i)
                y, uj = want_ch[n2], want_ch[n2 + 5:]
                rr = int.from_bytes(r_i, "little") % L
                assert slot(7) == rr
                rho = rr * rr % L if group > 1 else 1
                assert slot(13) == rho
                cp = rho * pow(y, pn - 1, L) % L
                for u in uj:
                    cp = cp * u * u % L
                ds, _, ss, opn = oracle.cloak_verify_prepare(c_i, n_in, n_out, p_i, r_i)
                assert opn == pn
                for name, got, want, cnt in (("static", st, ss, lay["n_static"]), ("dyn", dy, ds, lay["n_dyn"])):
                    assert len(want) == 32 * cnt
                    for j in range(cnt):
                        g = int.from_bytes(got[(i * cnt + j) * 32: (i * cnt + j + 1) * 32], "little")
                        w = int.from_bytes(want[32 * j: 32 * j + 32], "little") * cp % L
                        assert g == w, (n_in, n_out, i, name, j)
            # a malformed scalar, an identity point and a zeroed commitment: rejected in this mode too
            bad = bytearray(proofs)
            bad[1 + 32 * 11: 1 + 32 * 12] = L.to_bytes(32, "little")                         # t_x = l in transaction 0
            bad[plen + 1 + 32 * 6: plen + 1 + 32 * 7] = bytes(32)                             # T_1 = identity in transaction 1
            assert bits(v.verify_packed_gpu(n_in, n_out, batch, com, bytes(bad), plen, r), batch) == [0, 0, 1, 1, 1]
    finally:
        ctx.set_group_size(16)
        ctx.set_transcript_mode(0)
        v.close()


def test_benched_configuration_full_size_vs_oracle(ctx, oracle):
    """What bench.py times, as a parity test: 16-bit generator tables (34.5 GB), six batches in flight on forked
    contexts sharing the chip-filling streams, groups of 16, the 1024 DISTINCT golden proofs per batch (every step
    under fresh verifier randomness and its own corruptions), inputs resident in HBM -- every accept bit of
    every step against the oracle's full verifier; then one 4096-transaction batch."""
    from zkvm_amd.verifier import BulletproofGens, Verifier
    fix, n_in, n_out, plen = load_cloak_fixture()
    assert len(fix) == 1024 and len({p for _, p in fix}) == 1024
    gens = BulletproofGens(ctx, 256, table_bits=16)
    v = Verifier(ctx, gens)
    lanes = [ctx] + [ctx.fork() for _ in range(5)]
    ctx.set_group_size(16)
    w = 64 * (n_in + n_out)
    try:
        steps, batch = 12, 1024
        work = []
        for s in range(steps):
            coms, proofs = [], []
            for i in range(batch):
                com, proof = fix[(i + 31 * s) % 1024]
                if (i + s) % 97 == 5:
                    p = bytearray(proof); p[1 + 32 * (11 + i % 3) + (i % 31)] ^= 1 << (i % 8); proof = bytes(p)
                if (i + s) % 389 == 11:
                    cm = bytearray(com); cm[i % w] ^= 0x20; com = bytes(cm)
                coms.append(com); proofs.append(proof)
            r = hashlib.shake_256(b"benched %d" % s).digest(64 * batch)
            want = list(oracle.cloak_verify_batch(b"".join(coms), n_in, n_out, b"".join(proofs), plen, r, threads=16))
            assert 0 < want.count(0) < batch // 20
            work.append((ctx.to_device(b"".join(coms)), ctx.to_device(b"".join(proofs)), ctx.to_device(r), want))
        got = [None] * steps
        for s in range(steps + len(lanes)):
            c = lanes[s % len(lanes)]
            if s >= len(lanes):
                got[s - len(lanes)] = bits(c.verify_wait(), batch)
            if s < steps:
                d_com, d_pr, d_r, _ = work[s]
                v.submit_packed_gpu_dev(n_in, n_out, batch, d_com, d_pr, plen, d_r, ctx=c)
        for s in range(steps):
            assert got[s] == work[s][3], s
        for d_com, d_pr, d_r, _ in work:
            for d in (d_com, d_pr, d_r):
                ctx.free_device(d)
        big = 4096
        coms = [fix[(7 * i) % 1024][0] for i in range(big)]
        proofs = [fix[(7 * i) % 1024][1] for i in range(big)]
        for i in range(5, big, 211):
            p = bytearray(proofs[i]); p[-1 - (i % 64)] ^= 1; proofs[i] = bytes(p)
        r = hashlib.shake_256(b"benched big").digest(64 * big)
        want = list(oracle.cloak_verify_batch(b"".join(coms), n_in, n_out, b"".join(proofs), plen, r, threads=16))
        assert bits(v.verify_packed_gpu(n_in, n_out, big, b"".join(coms), b"".join(proofs), plen, r), big) == want
    finally:
        v.close()
        for c in lanes[1:]:
            c.close()
        gens.close()


@pytest.mark.parametrize("table_bits", [-1, 16])
def test_benched_arrangement_tickets_merged_into_device_batches_vs_oracle(ctx, oracle, table_bits):
    """EXACTLY what bench.py times with the driver's flags (--steps 20): 20 batches of 1024 transactions -- bench.py's own
    input construction (gpu_util.benched_step / benched_randomness: every batch its own rotation of the 1024 distinct
    golden proofs, its own corruptions, its own verifier randomness) -- queued by ONE zkgpu_verifier_submit_many_dev on a
    verifier with 5 lanes, merged into device batches of 10 240 transactions over the generator tables bench.py uses
    (table_bits = -1: the LIBRARY'S choice, 14-bit windows at a 128-byte row stride for these 514 points -- what the
    headline runs on since the end of round 4) and over 16-bit tables (34.5 GB), groups of 16, inputs resident in HBM: every accept bit of every ticket against the oracle's full verifier (and against
    the constructed expectation bench.py asserts); then the steady-state form: 40 more tickets, 64 in flight at most,
    submitted one by one as earlier ones are waited for."""
    from gpu_util import benched_randomness, benched_step
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    n_steps, batch, merge = 20, 1024, 10240
    gens = BulletproofGens(ctx, 256, table_bits=table_bits)
    assert gens.points.table_bits() == (14 if table_bits < 0 else table_bits)
    bv = BlockVerifier(ctx, gens, batches_in_flight=5)
    bv.set_merge(merge)
    assert 3 <= bv.lanes() <= 5              # (lanes whose streams would not run beside the others' are not kept)
    sets = []
    try:
        for s in range(n_steps + 40):
            txs, expected = benched_step(batch, 0, 64, s)
            r = benched_randomness(0, s, batch)
            n_in, n_out, plen = txs[0][0], txs[0][1], len(txs[0][3])
            com, proofs = b"".join(t[2] for t in txs), b"".join(t[3] for t in txs)
            if s < n_steps or s % 8 == 0:
                want = list(oracle.cloak_verify_batch(com, n_in, n_out, proofs, plen, r, threads=16))
                assert want == expected, s                          # the bench's constructed expectation IS the oracle's verdict
            t = [ctx.to_device(b) for b in (com, proofs, r)]        # (zkgpu_malloc + zkgpu_upload: device pointers)
            sets.append((t, expected))
        assert len({bytes(e) for _, e in sets[:n_steps]}) > 10      # the verdict pattern differs from step to step
        for _ in range(2):                                          # twice: cold and warm workspace
            tickets = bv.submit_many_dev(n_in, n_out, batch, [t[0] for t, _ in sets[:n_steps]], [t[1] for t, _ in sets[:n_steps]], plen,
                                         [t[2] for t, _ in sets[:n_steps]])
            for k, tk in enumerate(tickets):
                assert bits(bv.wait(tk), batch) == sets[k][1], k
        q = []
        for s in range(n_steps, n_steps + 40):
            if len(q) >= 16:
                k, tk = q.pop(0)
                assert bits(bv.wait(tk), batch) == sets[k][1], k
            t = sets[s][0]
            q.append((s, bv.submit_dev(n_in, n_out, batch, t[0], t[1], plen, t[2])))
        for k, tk in q:
            assert bits(bv.wait(tk), batch) == sets[k][1], k
    finally:
        bv.close()
        for t, _ in sets:
            for d in t:
                ctx.free_device(d)
        gens.close()


@pytest.mark.parametrize("table_bits", [-1, 12])
def test_tickets_from_host_memory_benched_steps_vs_oracle(ctx, oracle, table_bits):
    """zkgpu_verifier_submit / _submit_many: the benched step sets handed over in HOST memory -- one
    submit_many for the first 20, merged into device batches of 10 240 in pinned staging memory and copied to the lane's
    merge buffers by the verifier's copy stream -- every bit of every ticket against the oracle's verdicts; the caller's
    buffers overwritten right after the call (the library must have copied them); then one by one with up to 16 in flight,
    mixed with DEVICE tickets of other steps in the same queue; then the edge cases: a ticket larger than the merge target,
    a ticket of another shape in between, randomness from the OS (r_bytes = NULL), a proof length that is wrong for the
    shape (all bits zero, no error), a ticket waited for while its batch is still being formed."""
    import ctypes as C
    from gpu_util import benched_randomness, benched_step, mixed_block
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    n_steps, batch, merge = 20, 1024, 10240
    gens = BulletproofGens(ctx, 256, table_bits=table_bits)
    assert gens.points.table_bits() == (14 if table_bits < 0 else table_bits)      # -1: what bench.py's host_memory leg runs on
    bv = BlockVerifier(ctx, gens, batches_in_flight=5)
    bv.set_merge(merge)
    sets, dev = [], []
    try:
        for s in range(n_steps + 24):
            txs, expected = benched_step(batch, 0, 64, s)
            r = benched_randomness(0, s, batch)
            n_in, n_out, plen = txs[0][0], txs[0][1], len(txs[0][3])
            com, proofs = b"".join(t[2] for t in txs), b"".join(t[3] for t in txs)
            if s < 6 or s % 9 == 0:
                assert list(oracle.cloak_verify_batch(com, n_in, n_out, proofs, plen, r, threads=16)) == expected, s
            sets.append((com, proofs, r, expected))
        # mutable copies handed to the library and scribbled over as soon as the call is back
        bufs = [[C.create_string_buffer(x, len(x)) for x in st[:3]] for st in sets[:n_steps]]
        count = n_steps
        t = (C.c_uint64 * count)()
        ptrs = [(C.c_void_p * count)(*[C.addressof(b[k]) for b in bufs]) for k in range(3)]
        bv._check(bv.lib.zkgpu_verifier_submit_many(bv.h, n_in, n_out, count, batch, ptrs[0], ptrs[1], plen, ptrs[2], t))
        for b in bufs:
            for x in b:
                C.memset(x, 0xA5, len(x))
        for k in range(count):
            bm = C.create_string_buffer(batch // 8)
            bv._check(bv.lib.zkgpu_verifier_wait(bv.h, t[k], bm))
            assert bits(bm.raw, batch) == sets[k][3], k
        # one by one, 16 in flight, every third step as a DEVICE ticket in the same queue
        q = []
        for s in range(n_steps, n_steps + 24):
            if len(q) >= 16:
                k, tk = q.pop(0)
                assert bits(bv.wait(tk), batch) == sets[k][3], k
            com, proofs, r, _ = sets[s]
            if s % 3 == 0:
                d = [ctx.to_device(x) for x in (com, proofs, r)]
                dev.append(d)
                q.append((s, bv.submit_dev(n_in, n_out, batch, d[0], d[1], plen, d[2])))
            else:
                q.append((s, bv.submit(n_in, n_out, batch, com, proofs, plen, r)))
        for k, tk in q:
            assert bits(bv.wait(tk), batch) == sets[k][3], k
        # edge cases
        big_com, big_proofs, big_r = (b"".join(sets[s][j] for s in range(11)) for j in range(3))            # 11 264 > merge
        big_want = [b for s in range(11) for b in sets[s][3]]
        other = [tx for tx in mixed_block(60, seed=5, bad_every=5) if (tx[0], tx[1]) == (1, 2)]
        assert len(other) >= 4
        o_r = hashlib.shake_256(b"other shape").digest(64 * len(other))
        o_want = [int(oracle.cloak_verify(tx[2], 1, 2, tx[3], o_r[64 * i: 64 * i + 64])) for i, tx in enumerate(other)]
        assert 0 in o_want and 1 in o_want
        t_a = bv.submit(n_in, n_out, batch, *sets[0][:2], plen, sets[0][2])
        t_o = bv.submit(1, 2, len(other), b"".join(tx[2] for tx in other), b"".join(tx[3] for tx in other), len(other[0][3]), o_r)
        t_big = bv.submit(n_in, n_out, 11 * batch, big_com, big_proofs, plen, big_r)
        t_os = bv.submit(n_in, n_out, batch, *sets[1][:2], plen, None)                     # the OS's randomness
        t_len = bv.submit(n_in, n_out, 8, sets[2][0][: 8 * 256], sets[2][1][: 8 * (plen - 32)], plen - 32, sets[2][2][: 8 * 64])
        t_b = bv.submit(n_in, n_out, batch, *sets[3][:2], plen, sets[3][2])
        assert bits(bv.wait(t_b), batch) == sets[3][3]                                     # waited for first: its batch goes out as it is
        assert bits(bv.wait(t_len), 8) == [0] * 8
        assert bits(bv.wait(t_os), batch) == sets[1][3]                                    # (no verdict here depends on r)
        assert bits(bv.wait(t_big), 11 * batch) == big_want
        assert bits(bv.wait(t_o), len(other)) == o_want
        assert bits(bv.wait(t_a), batch) == sets[0][3]
        with pytest.raises(Exception):
            bv.wait(t_a)                                                                   # a ticket is waited for once
    finally:
        bv.close()
        for d in dev:
            for x in d:
                ctx.free_device(x)
        gens.close()


@pytest.mark.timeout(420, method="thread")
def test_host_memory_tickets_waited_for_newest_first_with_more_full_batches_than_lanes(ctx, oracle):
    """48 tickets of 1024 from host memory at a merge target of 4096 on 4 lanes are 12 full device
    batches -- more than lanes + 1 -- so that after the submit some FULL batches still stand in the forming list; the
    tickets are then waited for NEWEST FIRST.  host_dispatch used to return at the first older full batch it could find
    no lane for and never reached the one waited for: zkgpu_verifier_wait spun for ever holding the verifier's mutex
    (the timeout of this test is what would report it).  Every bit against the oracle's verdicts; then the same with the
    waits in a drawn order and device tickets of other steps in between."""
    import random
    from gpu_util import benched_randomness, benched_step
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    n, batch, merge = 48, 1024, 4096
    gens = BulletproofGens(ctx, 256, table_bits=12)
    bv = BlockVerifier(ctx, gens, batches_in_flight=4)
    bv.set_merge(merge)
    assert bv.lanes() <= 4
    sets, dev = [], []
    try:
        for s in range(n):
            txs, expected = benched_step(batch, 0, 64, 100 + s)
            r = benched_randomness(0, 100 + s, batch)
            n_in, n_out, plen = txs[0][0], txs[0][1], len(txs[0][3])
            com, proofs = b"".join(t[2] for t in txs), b"".join(t[3] for t in txs)
            if s % 12 == 0:
                assert list(oracle.cloak_verify_batch(com, n_in, n_out, proofs, plen, r, threads=16)) == expected, s
            sets.append((com, proofs, r, expected))
        tickets = [bv.submit(n_in, n_out, batch, c, p, plen, r) for c, p, r, _ in sets]
        for k in reversed(range(n)):
            assert bits(bv.wait(tickets[k]), batch) == sets[k][3], k
        rng = random.Random(5)
        tickets = []
        for k, (c, p, r, _) in enumerate(sets):
            if k % 5 == 2:
                d = [ctx.to_device(x) for x in (c, p, r)]
                dev.append(d)
                tickets.append(bv.submit_dev(n_in, n_out, batch, d[0], d[1], plen, d[2]))
            else:
                tickets.append(bv.submit(n_in, n_out, batch, c, p, plen, r))
        order = list(range(n))
        rng.shuffle(order)
        for k in order:
            assert bits(bv.wait(tickets[k]), batch) == sets[k][3], k
    finally:
        bv.close()
        for d in dev:
            for x in d:
                ctx.free_device(x)
        gens.close()


def test_msm_2p20_equals_committed_expected_value(ctx):
    """BASELINE.json configs[2] at full size against tests/golden/msm_2p20.json (the oracle's result, computed in
    the build container): inputs regenerated from SHAKE256 here, points mapped on the device -- whose 32 MiB of
    encodings must hash to the digest the oracle's from_uniform_bytes produced."""
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "msm_2p20.json")))
    n = gold["n"]
    sc, uniform = msm_2p20_inputs(n)
    assert hashlib.sha256(sc).hexdigest() == gold["scalars_sha256"]
    pts = ctx.hash_to_points(uniform)
    assert hashlib.sha256(pts).hexdigest() == gold["points_sha256"]
    for m, want in sorted((int(a), b) for a, b in gold["results"].items()):
        assert ctx.msm(sc[: 32 * m], pts[: 32 * m]).hex() == want, m


def test_failed_groups_are_located_and_every_pattern_of_bad_transactions_resolves(ctx, oracle):
    """Group checks with a failing group: one bad transaction per group (located by the index-weighted sum, checked
    alone, the others accepted through S1 - E_b), two and three bad ones in one group (no single culprit: re-checked
    one by one), bad ones at the first / last position, a bad one next to a transaction that is left out of the group
    (undecodable point, malformed proof), every transaction of a group bad -- always the oracle's per-transaction
    verdicts; and the ungrouped re-run (forced through the test hook) gives the same bits."""
    from zkvm_amd.verifier import BulletproofGens, Verifier
    fix, n_in, n_out, plen = load_cloak_fixture()
    gens = BulletproofGens(ctx, 256, table_bits=10)
    v = Verifier(ctx, gens)
    ctx.set_group_size(16)
    n = 16 * 12 + 5                                           # a ragged last group too
    coms = [bytearray(fix[i][0]) for i in range(n)]
    proofs = [bytearray(fix[i][1]) for i in range(n)]

    def bad_scalar(i): proofs[i][1 + 32 * 11 + (i % 29)] ^= 1 << (i % 8)          # t_x: only the MSM can tell
    def bad_a(i):
        a = (int.from_bytes(proofs[i][-64:-32], "little") + 1) % L
        proofs[i][-64:-32] = a.to_bytes(32, "little")
    def bad_point(i): coms[i][32:64] = bytes.fromhex("01" + "00" * 31)             # undecodable: known before the sums
    def malformed(i): proofs[i][1 + 32 * 12: 1 + 32 * 13] = L.to_bytes(32, "little")   # non-canonical scalar
    bad_scalar(16 * 0 + 7)                                    # group 0: one bad, in the middle
    bad_a(16 * 1 + 0)                                         # group 1: first position
    bad_scalar(16 * 2 + 15)                                   # group 2: last position
    bad_scalar(16 * 3 + 2); bad_a(16 * 3 + 9)                 # group 3: two bad
    bad_scalar(16 * 4 + 1); bad_scalar(16 * 4 + 2); bad_a(16 * 4 + 3)   # group 4: three
    bad_point(16 * 5 + 4)                                     # group 5: only a left-out one -> the group passes
    bad_point(16 * 6 + 4); bad_scalar(16 * 6 + 11)            # group 6: left-out + one culprit
    malformed(16 * 7 + 0); bad_a(16 * 7 + 15)                 # group 7: malformed + culprit
    for i in range(16):
        bad_scalar(16 * 8 + i)                                # group 8: all bad
    bad_scalar(16 * 12 + 4)                                   # the ragged group (5 transactions)
    r = hashlib.shake_256(b"locate").digest(64 * n)
    com_b, proof_b = b"".join(bytes(c) for c in coms), b"".join(bytes(p) for p in proofs)
    want = list(oracle.cloak_verify_batch(com_b, n_in, n_out, proof_b, plen, r, threads=16))
    assert want.count(0) == 1 + 1 + 1 + 2 + 3 + 1 + 2 + 2 + 16 + 1
    try:
        before = ctx.force_regroup(False)
        for horner in (1, 2, 0):                                # Horner chains per transaction / per group / automatic
            ctx.set_horner_mode(horner)
            for mode in (1, 2, 2, 3, 3):                        # re-check in full; locate (the default from 2048 per batch on); locating sums formed up front
                ctx.set_locate_mode(mode)
                assert bits(v.verify_packed_gpu(n_in, n_out, n, com_b, proof_b, plen, r), n) == want, (horner, mode)
        ctx.set_tail_mode(1)                                    # the tail's sums as launches of their own
        for mode in (1, 2, 3):
            ctx.set_locate_mode(mode)
            assert bits(v.verify_packed_gpu(n_in, n_out, n, com_b, proof_b, plen, r), n) == want, mode
        ctx.set_locate_mode(2)
        for parts in (1, 5, 64, 0):                             # lanes per (failed group, window) of the locating multiplication
            ctx.set_locate_parts(parts)
            assert bits(v.verify_packed_gpu(n_in, n_out, n, com_b, proof_b, plen, r), n) == want, parts
        ctx.set_tail_mode(0)
        assert ctx.force_regroup(True) == before               # nothing above needed the ungrouped re-run
        assert bits(v.verify_packed_gpu(n_in, n_out, n, com_b, proof_b, plen, r), n) == want
        assert ctx.force_regroup(False) == before + 1          # ... and here it was taken
        for group in (4, 64, 1):
            ctx.set_group_size(group)
            assert bits(v.verify_packed_gpu(n_in, n_out, n, com_b, proof_b, plen, r), n) == want, group
    finally:
        ctx.set_group_size(16)
        ctx.set_locate_mode(0)
        ctx.set_locate_parts(0)
        ctx.set_tail_mode(0)
        ctx.set_horner_mode(0)
        ctx.force_regroup(False)
        v.close()
        gens.close()


@pytest.mark.parametrize("kind,param", [(1, 32), (1, 64), (2, 2), (2, 5), (2, 9)])
def test_described_constraint_systems_on_device(ctx, oracle, kind, param):
    """zkgpu_r1cs_plan_create: a constraint system handed over as data -- a bare range proof
    (single phase) and a scalar shuffle (second-phase challenge, multipliers only in phase 2), written down in
    tests/gpu_util.py independently of the library and of the oracle -- verified with the transcript replay, the scalars
    and the multiscalar multiplications on the device: verdicts = the oracle's (its own gadget code, oracle/gadgets.c),
    incl. invalid witnesses and corrupted proofs; the device's challenges and every scalar of the equation = the
    oracle's (times c'); the host-prepared form (zkgpu_r1cs_verify_batch) agrees."""
    import random
    from gpu_util import GADGET_LABEL, describe_range, describe_shuffle
    from zkvm_amd.native import R1csDescription
    from zkvm_amd.verifier import BulletproofGens, R1csVerifier
    rng = random.Random(1000 * kind + param)
    m, n1, n, labels, cons = describe_range(param) if kind == 1 else describe_shuffle(param)
    desc = R1csDescription(GADGET_LABEL, m, n1, n, labels, cons)
    gens = BulletproofGens(ctx, 64, table_bits=8)
    v = R1csVerifier(ctx, gens, desc)
    ctx.set_group_size(1)
    try:
        info = v.info()
        batch = 21
        coms, proofs = [], []
        for i in range(batch):
            if kind == 1:
                values = [rng.randrange(1 << param)]
                if i == 4:
                    values = [(1 << param) + 3]                       # out of range: the proof exists but does not verify
            else:
                xs = [rng.randrange(L) for _ in range(param)]
                ys = xs[:]
                rng.shuffle(ys)
                if i == 4:
                    ys[0] = (ys[0] + 1) % L                           # not a permutation
                values = xs + ys
            rc, com, proof = oracle.gadget_prove(kind, param, values, hashlib.sha256(b"gadget %d %d %d" % (kind, param, i)).digest())
            assert rc == 0 and len(proof) == info["proof_len"]
            coms.append(bytearray(com)); proofs.append(bytearray(proof))
        proofs[7][1 + 32 * 12 + 5] ^= 0x10                            # t_x_blinding
        coms[9][3] ^= 1                                               # a commitment
        proofs[11][1 + 32 * 6: 1 + 32 * 7] = bytes(32)                # T_1 = identity: malformed
        proofs[13][0] = 3                                             # wire-format version
        r = hashlib.shake_256(b"gadget r %d %d" % (kind, param)).digest(64 * batch)
        com_b, proof_b = b"".join(bytes(c) for c in coms), b"".join(bytes(p) for p in proofs)
        want = [int(oracle.gadget_verify(kind, param, bytes(coms[i]), bytes(proofs[i]), r[64 * i: 64 * i + 64])) for i in range(batch)]
        assert want == [0 if i in (4, 7, 9, 11, 13) else 1 for i in range(batch)]
        assert bits(v.verify_gpu(batch, com_b, proof_b, info["proof_len"], r), batch) == want
        # the device's head, byte by byte (transactions checked alone: rho = 1)
        ch = ctx.debug_read("challenges", batch * info["slots"] * 32)
        st = ctx.debug_read("static_scalars", batch * info["n_static"] * 32)
        dy = ctx.debug_read("dyn_scalars", batch * info["n_dyn"] * 32)
        n2, k, pn = info["n_chal2"], info["k"], info["padded_n"]
        for i in (0, 1, 20):
            ds, _, ss, opn, want_ch = oracle.gadget_verify_prepare(kind, param, bytes(coms[i]), bytes(proofs[i]), r[64 * i: 64 * i + 64])
            assert opn == pn and len(want_ch) == n2 + 5 + k

            def slot(j):
                o = (i * info["slots"] + j) * 32
                return int.from_bytes(ch[o: o + 32], "little") * R260_INV % L
            assert [slot(14 + j) for j in range(n2)] + [slot(j) for j in range(5)] + [slot(14 + n2 + j) for j in range(k)] == want_ch
            cp = pow(want_ch[n2], pn - 1, L)
            for u in want_ch[n2 + 5:]:
                cp = cp * u * u % L
            for got, wantb, cnt in ((st, ss, info["n_static"]), (dy, ds, info["n_dyn"])):
                assert len(wantb) == 32 * cnt
                for j in range(cnt):
                    g = int.from_bytes(got[(i * cnt + j) * 32: (i * cnt + j + 1) * 32], "little")
                    assert g == int.from_bytes(wantb[32 * j: 32 * j + 32], "little") * cp % L, (i, j)
        assert bits(v.verify_host_prepared(batch, com_b, proof_b, info["proof_len"], r, host_threads=4), batch) == want
        ctx.set_group_size(16)
        assert bits(v.verify_gpu(batch, com_b, proof_b, info["proof_len"], r), batch) == want
        assert bits(v.verify_gpu(batch, com_b, proof_b, info["proof_len"]), batch) == want       # getrandom
    finally:
        ctx.set_group_size(16)
        v.close()
        gens.close()


@pytest.mark.parametrize("kind,param", [(3, 8), (2, 5), (1, 32)])
def test_described_prover_on_device_equals_oracle_and_verifies(ctx, oracle, kind, param):
    """zkgpu_r1cs_prove_batch (BASELINE.json configs[4]; kind 3 with 8 values = the 1032-constraint, 512-multiplier
    program): every commitment and proof byte equals the oracle's gadget prover on the same witness and seed, the
    device-side verifier (plan made from the same description) and the oracle accept them, and a statement with an
    invalid witness yields a proof both reject."""
    import random
    from gpu_util import GADGET_LABEL, describe_range, describe_ranges, describe_shuffle, gadget_witness
    from zkvm_amd.native import R1csDescription
    from zkvm_amd.verifier import BulletproofGens, R1csProver, R1csVerifier
    rng = random.Random(31 * kind + param)
    m, n1, n, labels, cons = describe_range(param) if kind == 1 else describe_shuffle(param) if kind == 2 else describe_ranges(param)
    desc = R1csDescription(GADGET_LABEL, m, n1, n, labels, cons)
    cap = 1
    while cap < max(n, 1):
        cap *= 2
    gens = BulletproofGens(ctx, cap, table_bits=8)
    batch = 9
    vals, givens, seeds, mult_def = [], [], [], None
    for i in range(batch):
        if kind == 1:
            values = [rng.randrange(1 << param)]
        elif kind == 3:
            values = [rng.randrange(1 << 64) for _ in range(param)]
        else:
            xs = [rng.randrange(L) for _ in range(param)]
            values = xs + sorted(xs)
        mult_def, given = gadget_witness(kind, param, values)
        if i == 5:                                                       # an inconsistent statement
            if kind == 2:
                values[-1] = (values[-1] + 1) % L
            else:
                values[0] += 1 << (param if kind == 1 else 64)           # the bits given are those of the value mod 2^bits
        vals.append(values); givens.append(given); seeds.append(hashlib.sha256(b"dev desc prover %d %d %d" % (kind, param, i)).digest())
    try:
        for mode in (1, 0, 16 + 3):                             # host threads in lockstep; the whole proof on the device; the same
            ctx.set_prover_mode(mode)                           # as three slices in flight on streams of their own (round 5)
            coms, proofs = R1csProver(ctx, gens, desc, mult_def, host_threads=8).prove(vals, givens, seeds)
            for i in range(batch):
                rc, want_com, want_proof = oracle.gadget_prove(kind, param, vals[i], seeds[i])
                assert rc == 0 and coms[i] == want_com and proofs[i] == want_proof, (mode, i)
    finally:
        ctx.set_prover_mode(0)
    r = hashlib.shake_256(b"desc prover r").digest(64 * batch)
    want = [int(oracle.gadget_verify(kind, param, coms[i], proofs[i], r[64 * i: 64 * i + 64])) for i in range(batch)]
    assert want == [0 if i == 5 else 1 for i in range(batch)]
    v = R1csVerifier(ctx, gens, desc)
    try:
        assert bits(v.verify_gpu(batch, b"".join(coms), b"".join(proofs), len(proofs[0]), r), batch) == want
    finally:
        v.close()
        gens.close()


def test_tickets_merge_small_batches_and_return_each_its_own_bitmap(ctx, oracle):
    """zkgpu_verifier_submit_dev / zkgpu_verifier_wait: fifteen batches of two shapes and ragged sizes queued as tickets
    on a verifier with three contexts, merged into device batches of at most 1500 transactions; every ticket gets the
    oracle's verdicts for ITS transactions, whatever the order they are waited for in, also with a block verification
    in between, and tickets of a shape the generators cannot serve are rejected one by one."""
    import random
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    fix = load_mixed_fixture()
    gens = BulletproofGens(ctx, 256, table_bits=9)
    bv = BlockVerifier(ctx, gens, batches_in_flight=3)
    bv.set_merge(1500)
    rng = random.Random(12)
    try:
        jobs = []
        for j in range(15):
            shape = [(2, 2), (1, 2), (2, 2), (4, 4)][j % 4] if j != 7 else (2, 2)
            n = rng.choice([1, 37, 256, 400, 640])
            recs = [fix[shape][rng.randrange(32)] for _ in range(n)]
            coms = [bytearray(c) for c, _ in recs]
            proofs = [bytearray(p) for _, p in recs]
            for i in range(0, n, 29):
                proofs[i][1 + 32 * 11 + (i % 20)] ^= 1 << (j % 8)
            r = hashlib.shake_256(b"ticket %d" % j).digest(64 * n)
            plen = len(proofs[0])
            com_b, proof_b = b"".join(bytes(c) for c in coms), b"".join(bytes(p) for p in proofs)
            if shape == (4, 4):
                want = [0] * n                                          # 512 multipliers, 256 generators
            else:
                want = list(oracle.cloak_verify_batch(com_b, shape[0], shape[1], proof_b, plen, r, threads=8))
            jobs.append((shape, n, ctx.to_device(com_b), ctx.to_device(proof_b), ctx.to_device(r), plen, want))
        tickets = [bv.submit_dev(s[0], s[1], n, dc, dp, plen, dr) for (s, n, dc, dp, dr, plen, _) in jobs[:10]]
        order = list(range(10))
        rng.shuffle(order)
        for k in order[:4]:
            assert bits(bv.wait(tickets[k]), jobs[k][1]) == jobs[k][6], k
        block = mixed_block(90, seed=3, bad_every=11)
        block = [t for t in block if (t[0], t[1]) != (4, 4)]
        rb = hashlib.shake_256(b"ticket block").digest(64 * len(block))
        assert bits(bv.verify(_cloak(block), rb), len(block)) == oracle_block_bits(oracle, block, rb)     # finishes the tickets in flight first
        tickets += [bv.submit_dev(s[0], s[1], n, dc, dp, plen, dr) for (s, n, dc, dp, dr, plen, _) in jobs[10:]]
        for k in order[4:] + list(range(10, 15)):
            assert bits(bv.wait(tickets[k]), jobs[k][1]) == jobs[k][6], k
        # zkgpu_verifier_submit_many_dev: 20 equal batches in one call (17 merged by the first gather launch, 3 by the next)
        shape, n, dc, dp, dr, plen, want = next(j for j in jobs if j[0] == (2, 2) and j[1] == 37) if any(j[0] == (2, 2) and j[1] == 37 for j in jobs) else jobs[0]
        bv.set_merge(20 * n)
        many = bv.submit_many_dev(shape[0], shape[1], n, [dc] * 20, [dp] * 20, plen, [dr] * 20)
        assert len(set(many)) == 20
        for t in reversed(many):
            assert bits(bv.wait(t), n) == want
        for (_, _, dc, dp, dr, _, _) in jobs:
            for d in (dc, dp, dr):
                ctx.free_device(d)
    finally:
        bv.close()
        gens.close()


@pytest.mark.parametrize("seed,shape", [(1, None), (2, None), (3, None), (4, None), (5, (1, 0, 0)), (6, (2, 1, 0)), (7, (3, 0, 2)),
                                        (8, (4, 8, 4)), (9, (2, 3, 1)), (10, (3, 13, 3))])
def test_random_described_systems_all_provers_and_verifiers_agree(ctx, seed, shape):
    """Random constraint systems handed over as data (no code anywhere knows them): the device prover and the lockstep
    prover produce byte-identical proofs, the device verifier and the host-prepared verifier both accept them and both
    reject a proof made for other values -- four implementations of the R1CS layer against each other."""
    import random
    from zkvm_amd.native import R1csDescription
    from zkvm_amd.verifier import BulletproofGens, R1csProver, R1csVerifier
    rng = random.Random(9000 + seed)
    m, n1, n2 = shape if shape else (rng.randrange(1, 5), rng.randrange(0, 9), rng.randrange(0, 5) if seed != 2 else 0)
    n_chal = rng.randrange(1, 3) if n2 else 0
    (m, n1, n, labels, cons), mult_def, values, given = random_system(rng, m, n1, n2, n_chal)
    desc = R1csDescription(b"random system", m, n1, n, labels, cons)
    gens = BulletproofGens(ctx, 16, table_bits=8)
    batch = 5
    vals, givens, seeds = [values] * batch, [given] * batch, [hashlib.sha256(b"rs %d %d" % (seed, i)).digest() for i in range(batch)]
    try:
        out = {}
        for mode in (1, 0):
            ctx.set_prover_mode(mode)
            out[mode] = R1csProver(ctx, gens, desc, mult_def, host_threads=2).prove(vals, givens, seeds)
        assert out[0] == out[1]
        coms, proofs = out[0]
        assert len(set(proofs)) == batch                            # different seeds, different proofs
        bad = bytearray(proofs[2]); bad[1 + 32 * 11 + 3] ^= 1       # t_x
        proofs_t = list(proofs); proofs_t[2] = bytes(bad)
        coms_t = list(coms); coms_t[4] = coms[4][:-32] + coms[3][:32] if m > 1 else coms[4]
        r = hashlib.shake_256(b"random system r").digest(64 * batch)
        want = [1, 1, 0, 1, 0 if (m > 1 and coms_t[4] != coms[4]) else 1]
        v = R1csVerifier(ctx, gens, desc)
        try:
            plen = len(proofs[0])
            assert bits(v.verify_gpu(batch, b"".join(coms), b"".join(proofs), plen, r), batch) == [1] * batch
            assert bits(v.verify_gpu(batch, b"".join(coms_t), b"".join(proofs_t), plen, r), batch) == want
            assert bits(v.verify_host_prepared(batch, b"".join(coms_t), b"".join(proofs_t), plen, r), batch) == want
        finally:
            v.close()
    finally:
        ctx.set_prover_mode(0)
        gens.close()


@pytest.mark.parametrize("seed,merge,lanes", [(11, 3000, 4), (12, 10240, 5)])
def test_tickets_of_drawn_sizes_from_host_and_device_memory_waited_for_in_a_drawn_order(seed, merge, lanes):
    """tools/ticket_soak.py as a test: 250 tickets of drawn sizes (1 .. 1024 transactions), seven in ten from HOST memory (staged,
    copied to the HBM twin of their staging area ticket by ticket, the area held until its device batch is collected) and three
    in ten from device memory, in ONE queue, waited for in a drawn order with a drawn number in flight -- every verdict against
    the constructed expectation (a fresh process: the verifier's areas, lanes and queue start empty)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ticket_soak.py"), "250", str(seed), str(merge), str(lanes)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ticket soak ok: 250 tickets" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_ticket_bursts_leave_in_equal_device_batches_and_the_default_target_is_the_tuned_one(ctx, oracle):
    """Round 6: what is queued of one shape leaves in round(queued / target) device batches of EQUAL
    size, all of them in the call that decided the cut -- bursts of 16, 20, 24, 37 tickets at a target of ten tickets are
    2, 2, 2, 4 device batches (8 + 8, 10 + 10, 12 + 12, 10 + 9 + 9 + 9), counted by the k_batch_init launches on the lanes;
    half a target beyond a multiple makes one batch more (24 -> 2, 25 -> 3); every bit of every ticket against the
    oracle.  And a verifier nobody called set_merge on cuts at 10 240 transactions: 21 tickets of 1024 -> two batches."""
    from gpu_util import benched_randomness, benched_step
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    small, merge = 48, 480                                       # tickets of 48 transactions, ten to a target
    gens = BulletproofGens(ctx, 256, table_bits=10)
    sets = []
    for s in range(37):
        txs, expected = benched_step(small, 0, 16, s)
        r = benched_randomness(0, s, small)
        n_in, n_out, plen = txs[0][0], txs[0][1], len(txs[0][3])
        com, proofs = b"".join(t[2] for t in txs), b"".join(t[3] for t in txs)
        assert list(oracle.cloak_verify_batch(com, n_in, n_out, proofs, plen, r, threads=16)) == expected, s
        sets.append(([ctx.to_device(b) for b in (com, proofs, r)], expected))

    def device_batches(bv, run):
        lanes = [bv.lane(i) for i in range(bv.lanes())]
        for c in lanes:
            c.profile_reset()
            c.profile(True)
        run()
        n = 0
        for c in lanes:
            c.profile(False)
            n += int(c.profile_read().get("k_batch_init", (0, 0.0))[0])
        return n

    bv = BlockVerifier(ctx, gens, batches_in_flight=5)
    bv.set_merge(merge)
    try:
        for burst, want_batches in ((16, 2), (20, 2), (24, 2), (37, 4), (25, 3), (4, 1)):
            def run():
                tickets = bv.submit_many_dev(n_in, n_out, small, [t[0] for t, _ in sets[:burst]], [t[1] for t, _ in sets[:burst]], plen,
                                             [t[2] for t, _ in sets[:burst]])
                for k, tk in enumerate(tickets):
                    assert bits(bv.wait(tk), small) == sets[k][1], (burst, k)
            assert device_batches(bv, run) == want_batches, burst
    finally:
        bv.close()
    # the library's own target, 1024-transaction tickets (one step's inputs repeated: the verdicts repeat with them)
    big = 1024
    txs, expected = benched_step(big, 0, 64, 0)
    r = benched_randomness(0, 0, big)
    d = [ctx.to_device(b) for b in (b"".join(t[2] for t in txs), b"".join(t[3] for t in txs), r)]
    bv = BlockVerifier(ctx, gens, batches_in_flight=5)           # (no set_merge)
    try:
        def run21():
            tickets = bv.submit_many_dev(n_in, n_out, big, [d[0]] * 21, [d[1]] * 21, plen, [d[2]] * 21)
            for tk in tickets:
                assert bits(bv.wait(tk), big) == expected
        assert device_batches(bv, run21) == 2                    # 21 504 transactions: round(2.1) = 2 batches (11 + 10 tickets)
    finally:
        bv.close()
        for x in d:
            ctx.free_device(x)
        for t, _ in sets:
            for x in t:
                ctx.free_device(x)
        gens.close()
