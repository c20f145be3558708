"""Fail-closed, EXECUTED on the device paths (SURVEY.md sec 5 "fail-closed", sec 8(b) the
zkgpu_verify_batch convention: "on any device error returns nonzero and zeroes the bitmap").

The hook zkgpu_debug_fail_after(n) (csrc/fault_gate.hpp) makes the n-th HIP runtime call the library makes -- an
allocation, a copy, an event, a stream wait, a synchronisation, the hipGetLastError that collects a launch -- report an
error without being made.  Each test first counts the runtime calls N of a clean run of its scenario, then runs the
scenario N times with call 1, 2, ... N failing (sampled beyond a few hundred), and after EVERY faulty run once more
clean.  What must hold for every output of every faulty run:

    status != OK  ==>  the output is all zero          (an error is never an accept)
    status == OK  ==>  the output is the right one     (a fault that was absorbed -- a retry, an unchecked profiling
                                                        event -- must not have changed a verdict)

nothing may hang (pytest timeout), and the clean run after it must be right on the SAME verifier / context: lanes,
staging areas and workspaces come back.  "Right" is the oracle's verdict (the scenarios' expectations are compared with
the oracle's full verifier here, on the same bytes); for the prover, the oracle prover's bytes.
"""
import ctypes as C
import hashlib

import pytest

from gpu_util import bits, mixed_block, oracle_block_bits

pytestmark = pytest.mark.gpu
OK, EHIP = 0, -3


@pytest.fixture(scope="module")
def ctx():
    from zkvm_amd import Context
    c = Context(0)
    yield c
    c.lib.zkgpu_debug_fail_after(c.h, 0, None)
    c.close()


def _arm(ctx, n):
    ctx.lib.zkgpu_debug_fail_after(ctx.h, n, None)


def _disarm(ctx):
    """-> (runtime calls seen since arming, calls answered 'failed')"""
    fired = C.c_longlong(0)
    seen = ctx.lib.zkgpu_debug_fail_after(ctx.h, 0, C.byref(fired))
    return int(seen), int(fired.value)


def _plan(n_calls, dense=160, sampled=120):
    """which calls fail: every one of the first `dense`, then `sampled` spread evenly over the rest, and the last three"""
    pts = set(range(1, min(n_calls, dense) + 1))
    if n_calls > dense:
        step = max(1, (n_calls - dense) // sampled)
        pts.update(range(dense + 1, n_calls + 1, step))
        pts.update(range(max(1, n_calls - 2), n_calls + 1))
    return sorted(pts)


def _why(ctx, scenario):
    """the library's own account of the last error (context and, when the scenario names one, verifier)"""
    txt = [ctx.lib.zkgpu_last_error(ctx.h).decode()]
    v = getattr(scenario, "verifier", None)
    if v is not None and v.h:
        txt.append(ctx.lib.zkgpu_verifier_last_error(v.h).decode())
    return txt


def _faulty_run(ctx, scenario, check, label, n):
    """one run with the n-th call failing (n < 0: and every call after it), then one clean run on the same objects;
    -> (fired, surfaced)"""
    _arm(ctx, n)
    got = scenario()
    _, fired = _disarm(ctx)
    check(got, n)
    surfaced = any(st != OK for st, _, _ in got)
    assert fired or not surfaced, (label, n, "an error without a fault")
    again = scenario()                       # the same objects, clean: lanes, staging areas, workspaces came back
    check(again, None)
    assert all(st == OK for st, _, _ in again), (label, "after fault", n, [st for st, _, _ in again], _why(ctx, scenario))
    return fired, surfaced


def _sweep(ctx, make, check, label, cold_samples=24):
    """make() -> (scenario, close): fresh objects (verifier, tables ...); scenario() -> [(status, output, expected)].
    WARM sweep: on one set of objects whose workspaces exist, every runtime call of a run fails in turn, then a lost
    device at three depths.  COLD sweep: fresh objects per fault, so that the calls of a FIRST run -- workspace, staging
    and merge-buffer allocations, stream and event creation -- fail too (a sample: every fresh set costs a table build)."""
    scenario, close = make()
    try:
        _arm(ctx, 1 << 60)                   # counts, never fires
        clean = scenario()
        n_cold, fired = _disarm(ctx)
        assert fired == 0
        check(clean, None)
        assert all(st == OK for st, _, _ in clean), (label, [st for st, _, _ in clean])
        _arm(ctx, 1 << 60)
        check(scenario(), None)
        n_warm, _ = _disarm(ctx)
        assert 10 <= n_warm <= n_cold, (label, n_warm, n_cold)
        surfaced = absorbed = 0
        for n in _plan(n_warm):
            fired, err = _faulty_run(ctx, scenario, check, label, n)
            surfaced += bool(fired and err)
            absorbed += bool(fired and not err)
        for n in (1, max(2, n_warm // 3), max(3, n_warm // 2)):        # a device that is gone: every call from the n-th on fails
            fired, err = _faulty_run(ctx, scenario, check, label, -n)
            assert fired >= 1 and err, (label, "lost device at call", n)
    finally:
        _disarm(ctx)
        close()
    cold = 0
    extra = n_cold - n_warm
    pts = sorted(set(range(1, n_cold + 1, max(1, n_cold // cold_samples)))) if extra > 0 else []
    for n in pts:
        scenario, close = make()
        try:
            fired, err = _faulty_run(ctx, scenario, check, label + " (cold)", n)
            cold += bool(fired and err)
        finally:
            _disarm(ctx)
            close()
    assert surfaced >= len(_plan(n_warm)) // 2, (label, surfaced, absorbed, n_warm)
    print("%s: %d runtime calls in a first run, %d in a warm one; warm sweep: %d faults surfaced as errors, %d absorbed without a "
          "wrong bit; cold sweep: %d of %d sampled faults surfaced" % (label, n_cold, n_warm, surfaced, absorbed, cold, len(pts)))
    return n_cold, n_warm, surfaced, absorbed


def _check_outputs(label):
    def check(outs, n):
        for k, (st, out, want) in enumerate(outs):
            if st == OK:
                assert out == want, (label, "fault at call", n, "output", k, "status OK but wrong bytes")
            else:
                assert out == bytes(len(out)), (label, "fault at call", n, "output", k, "status", st, "but nonzero bytes")
    return check


def _benched_sets(ctx, oracle, n_steps, batch, compare=2):
    from gpu_util import benched_randomness, benched_step
    sets = []
    for s in range(n_steps):
        txs, expected = benched_step(batch, 0, 64, s)
        r = benched_randomness(0, s, batch)
        n_in, n_out, plen = txs[0][0], txs[0][1], len(txs[0][3])
        com, proofs = b"".join(t[2] for t in txs), b"".join(t[3] for t in txs)
        if s < compare:
            assert list(oracle.cloak_verify_batch(com, n_in, n_out, proofs, plen, r, threads=16)) == expected, s
        bm = bytearray(batch // 8)
        for i, b in enumerate(expected):
            bm[i // 8] |= b << (i % 8)
        sets.append((com, proofs, r, bytes(bm)))
    return sets, n_in, n_out, plen


@pytest.mark.timeout(1500, method="thread")
def test_device_tickets_merged_into_two_device_batches_fail_closed_at_every_runtime_call(ctx, oracle):
    """(a) twenty 1024-transaction tickets queued by ONE zkgpu_verifier_submit_many_dev, merged into two device batches of
    10 240 on a five-lane verifier (bench.py's arrangement): the merge kernel, the three pieces of each device batch
    (front / decoding / back: session.hpp flush_backs and its back-half failure branch), the waits."""
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    n_steps, batch = 20, 1024
    sets, n_in, n_out, plen = _benched_sets(ctx, oracle, n_steps, batch)
    gens = BulletproofGens(ctx, 256, table_bits=12)
    dev = [[ctx.to_device(x) for x in st[:3]] for st in sets]
    lib = ctx.lib
    arr = lambda k: (C.c_void_p * n_steps)(*[d[k] for d in dev])            # noqa: E731

    def make():
        bv = BlockVerifier(ctx, gens, batches_in_flight=5)
        bv.set_merge(10240)

        def scenario():
            t = (C.c_uint64 * n_steps)()
            rc = lib.zkgpu_verifier_submit_many_dev(bv.h, n_in, n_out, n_steps, batch, arr(0), arr(1), plen, arr(2), t)
            assert rc == OK                     # (queueing itself cannot fail: a failed launch is the tickets' status)
            outs = []
            for k in range(n_steps):
                bm = C.create_string_buffer(b"\xff" * (batch // 8), batch // 8)
                st = lib.zkgpu_verifier_wait(bv.h, t[k], bm)
                outs.append((st, bm.raw, sets[k][3]))
            return outs
        scenario.verifier = bv
        return scenario, bv.close

    try:
        _sweep(ctx, make, _check_outputs("device tickets"), "device tickets")
    finally:
        _disarm(ctx)
        for d in dev:
            for x in d:
                ctx.free_device(x)
        gens.close()


@pytest.mark.timeout(1500, method="thread")
def test_host_memory_tickets_fail_closed_and_their_staging_areas_come_back(ctx, oracle):
    """(b) the same steps handed over in HOST memory (zkgpu_verifier_submit_many: pinned staging areas with HBM twins, the
    copy stream, host_launch / host_batch_fail), merge target 4096 so that a run forms three device batches and a partly
    filled one that only the wait sends out.  A staging area that stayed 'taken' after a fault would starve the clean
    runs that follow every faulty one (hundreds of them on 2 x lanes + 2 areas): the test would hang, not pass."""
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    n_steps, batch = 14, 1024
    sets, n_in, n_out, plen = _benched_sets(ctx, oracle, n_steps, batch)
    gens = BulletproofGens(ctx, 256, table_bits=12)
    lib = ctx.lib
    ptrs = [(C.c_char_p * n_steps)(*[st[k] for st in sets]) for k in range(3)]

    def make():
        bv = BlockVerifier(ctx, gens, batches_in_flight=3)
        bv.set_merge(4096)

        def scenario():
            t = (C.c_uint64 * n_steps)()
            rc = lib.zkgpu_verifier_submit_many(bv.h, n_in, n_out, n_steps, batch, ptrs[0], ptrs[1], plen, ptrs[2], t)
            assert rc == OK
            outs = []
            for k in reversed(range(n_steps)):   # newest first: the partly filled batch goes out by the wait
                bm = C.create_string_buffer(b"\xff" * (batch // 8), batch // 8)
                st = lib.zkgpu_verifier_wait(bv.h, t[k], bm)
                outs.append((st, bm.raw, sets[k][3]))
            return outs
        scenario.verifier = bv
        return scenario, bv.close

    try:
        _sweep(ctx, make, _check_outputs("host tickets"), "host tickets")
    finally:
        _disarm(ctx)
        gens.close()


@pytest.mark.timeout(1500, method="thread")
def test_sharded_verification_in_a_mocked_world_fails_closed_and_still_enters_the_collective(ctx, oracle):
    """(c) zkgpu_verifier_verify_sharded as rank 1 of a mocked world of two (zkgpu_debug_comm_mock: the collective's
    function table replaced, every buffer, copy, stream and frame the product's own): a fault anywhere in this rank's
    block -- staging, upload, its device batches, the exchange's copies -- gives a nonzero status and an all-zero bitmap
    HERE, and the collective is entered all the same whenever the fault came before it (a rank that skipped it would leave
    its peers waiting for ever: the mock counts the all-gathers it served)."""
    from test_gpu_block import _cloak, _slot
    from zkvm_amd.native import Comm, shard_cuts
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens, _marshal_block
    world, rank = 2, 1
    lib = ctx.lib
    txs = [t for t in mixed_block(90, seed=77, bad_every=7) if (t[0], t[1]) != (4, 4)]
    n = len(txs)
    r = hashlib.shake_256(b"faulty world").digest(64 * n)
    want = oracle_block_bits(oracle, txs, r)
    assert 0 in want and 1 in want
    cuts = shard_cuts([(t[0], t[1]) for t in txs], world)
    whole = bytearray((n + 7) // 8)
    for i, v in enumerate(want):
        whole[i // 8] |= v << (i % 8)

    def shard_bitmap(k):
        b = bytearray((cuts[k + 1] - cuts[k] + 7) // 8)
        for j, i in enumerate(range(cuts[k], cuts[k + 1])):
            b[j // 8] |= want[i] << (j % 8)
        return bytes(b)

    peers = [_slot(cuts, k, shard_bitmap(k)) for k in range(world)]
    peers[rank] = b"\xff" * len(peers[0])
    blob = b"".join(peers)
    assert lib.zkgpu_debug_comm_mock(ctx.h, world, blob, len(peers[0])) >= 0
    gens = BulletproofGens(ctx, 256, table_bits=8)
    n_in, n_out, com, proofs, po = _marshal_block(_cloak(txs), r)
    gathers = []

    def make():
        bv = BlockVerifier(ctx, gens, batches_in_flight=2)
        comm = Comm(ctx, rank, world, bytes(128))

        def scenario():
            before = lib.zkgpu_debug_comm_mock(ctx.h, -1, None, 0)          # (world < 0: a query)
            bm = C.create_string_buffer(b"\xff" * len(whole), len(whole))
            st = lib.zkgpu_verifier_verify_sharded(bv.h, comm.h, n, n_in, n_out, com, proofs, po, r, bm)
            gathers.append(lib.zkgpu_debug_comm_mock(ctx.h, -1, None, 0) - before)
            return [(st, bm.raw, bytes(whole))]

        def close():
            comm.close()
            bv.close()
        scenario.verifier = bv
        return scenario, close

    try:
        _sweep(ctx, make, _check_outputs("sharded"), "sharded")
        # every run entered the collective exactly once, unless the fault hit the exchange's own calls after this rank's
        # contribution could no longer be sent (then zero: the error is this rank's and is reported here)
        assert set(gathers) <= {0, 1} and gathers.count(1) >= len(gathers) * 0.9, (gathers.count(0), gathers.count(1))
        print("sharded: runs that entered the collective: %d of %d" % (gathers.count(1), len(gathers)))
    finally:
        _disarm(ctx)
        lib.zkgpu_debug_comm_mock(ctx.h, 0, None, 0)
        gens.close()


@pytest.mark.timeout(1500, method="thread")
def test_sliced_prover_call_fails_closed_in_every_slice(ctx, oracle):
    """(d) zkgpu_cloak_prove_batch cut into three slices (zkgpu_set_prover_mode 16 + 3: three contexts, three threads,
    one shared table set -- zkgpu.hip run_sliced): a fault in ANY slice fails the whole call, every commitment and every
    proof byte of EVERY slice is zero (a caller must never publish half a call), the threads are joined, and the next
    call gives the oracle prover's bytes again."""
    from zkvm_amd import Context
    from zkvm_amd.verifier import BulletproofGens
    lib = ctx.lib
    batch, n_in, n_out, nv = 9, 2, 2, 4
    qs = [[5 + i, 9, 4 + i, 10] for i in range(batch)]
    fl = bytes([3]) + bytes(31)
    seeds = [hashlib.sha256(b"faulty prover %d" % i).digest() for i in range(batch)]
    stride = 1 + 32 * (16 + 2 * 16)
    want_com, want_proofs = b"", b""
    for i in range(batch):
        rc, c, p, _ = oracle.cloak_prove(qs[i], [fl] * nv, n_in, n_out, seeds[i])
        assert rc == 0
        want_com += c
        want_proofs += p + bytes(stride - len(p))
    gens = BulletproofGens(ctx, 256, table_bits=8)
    qa = (C.c_uint64 * (batch * nv))(*[q for row in qs for q in row])
    flb, sd = fl * (nv * batch), b"".join(seeds)

    def make():
        pc = Context(0)                       # a context of its own per set: the slices' helper contexts are made by its first call
        pc.set_prover_mode(16 + 3)

        def scenario():
            com = C.create_string_buffer(b"\xff" * (64 * nv * batch), 64 * nv * batch)
            proofs = C.create_string_buffer(b"\xff" * (stride * batch), stride * batch)
            plen = C.c_size_t(0)
            st = lib.zkgpu_cloak_prove_batch(pc.h, gens.points.h, gens.gens_capacity, batch, n_in, n_out, qa, flb, sd, 3, com, proofs, stride, C.byref(plen))
            got = proofs.raw
            if st == OK:                      # (the stride's padding beyond proof_len is the caller's: compare what was written)
                got = b"".join(got[stride * i: stride * i + plen.value] + bytes(stride - plen.value) for i in range(batch))
            return [(st, com.raw, want_com), (st, got, want_proofs)]
        return scenario, pc.close

    try:
        _sweep(ctx, make, _check_outputs("sliced prover"), "sliced prover")
    finally:
        _disarm(ctx)
        gens.close()


@pytest.mark.timeout(1500, method="thread")
def test_the_four_entry_points_of_the_survey_fail_closed(ctx, oracle):
    """(e) SURVEY.md sec 8(b): zkgpu_msm (out = 32 zero bytes and a nonzero status on a device error: the partition sort, the
    decompression beside it on the second stream, the fat / heavy bins) and zkgpu_verify_batch (bitmap zeroed) on a context of
    their own; zkgpu_init / zkgpu_destroy are what every fresh set of objects of the sweep goes through."""
    from gpu_util import points, scalars
    from zkvm_amd import Context
    lib = ctx.lib
    n = 40000                                              # (the partition-sort path: >= 32 768 terms)
    sc, pt = scalars("fault msm", n), points(oracle, "fault msm", n, distinct=211)
    rc, want, _ = oracle.msm(sc, pt)
    assert rc == 0
    m = 300
    bsc, bpt, offs, expect = b"", b"", [0], []
    for i in range(m):
        k = int.from_bytes(sc[32 * i: 32 * i + 32], "little")
        p = pt[32 * i: 32 * i + 32]
        good = i % 7 != 3
        bsc += k.to_bytes(32, "little") + ((L_ORDER - k + (0 if good else 1)) % L_ORDER).to_bytes(32, "little")
        bpt += p + p
        offs.append(offs[-1] + 2)
        expect.append(1 if good else 0)
    want_bm = bytearray((m + 7) // 8)
    for i, b in enumerate(expect):
        want_bm[i // 8] |= b << (i % 8)
    assert oracle.verify_batch(bsc, bpt, offs) == bytes(want_bm)
    c_offs = (C.c_uint64 * (m + 1))(*offs)

    def make():
        c2 = Context(0)

        def scenario():
            out = C.create_string_buffer(b"\xff" * 32, 32)
            bad = C.c_size_t(0)
            st = lib.zkgpu_msm(c2.h, sc, pt, n, out, C.byref(bad))
            bm = C.create_string_buffer(b"\xff" * len(want_bm), len(want_bm))
            st2 = lib.zkgpu_verify_batch(c2.h, bsc, bpt, c_offs, m, bm)
            return [(st, out.raw, want), (st2, bm.raw, bytes(want_bm))]
        return scenario, c2.close

    _sweep(ctx, make, _check_outputs("msm + verify_batch"), "msm + verify_batch", cold_samples=12)


L_ORDER = 2**252 + 27742317777372353535851937790883648493


@pytest.mark.timeout(1500, method="thread")
def test_serialized_transaction_calls_fail_closed_synchronous_and_in_flight(ctx, oracle):
    """(f) zkgpu_tx_verify_batch and two zkgpu_tx_verify_submit calls in flight on one verifier (the engine thread's rounds):
    staging arenas, the key and signature stages on their own contexts, the proofs as blocks over the ticket queue.  Both
    outputs fail closed: a nonzero status leaves every accept bit 0 and NO status byte 0 ("accepted"); with status OK bits and
    status bytes are the oracle's (the transactions' expected verdicts are held against the oracle's Tx::verify here)."""
    from gpu_util import built_transactions
    from zkvm_amd.verifier import BlockVerifier, BulletproofGens
    lib = ctx.lib
    n = 96
    txs, expected = built_transactions(n, call=7, bad_every=8)
    r = hashlib.shake_256(b"faulty tx").digest(64)
    assert [1 if oracle.tx_verify(t, r) == 0 else 0 for t in txs[:24]] == list(expected[:24]) and 0 in expected and 1 in expected
    want_bm = bytearray((n + 7) // 8)
    for i, b in enumerate(expected):
        want_bm[i // 8] |= b << (i % 8)
    want_st = bytes(0 if b else 1 for b in expected)
    blob = b"".join(txs)
    offs = (C.c_uint64 * (n + 1))(*([0] + [sum(len(t) for t in txs[:i + 1]) for i in range(n)]))
    half = n // 2
    blob_a, blob_b = b"".join(txs[:half]), b"".join(txs[half:])
    offs_a = (C.c_uint64 * (half + 1))(*([0] + [sum(len(t) for t in txs[:i + 1]) for i in range(half)]))
    offs_b = (C.c_uint64 * (n - half + 1))(*([0] + [sum(len(t) for t in txs[half:half + i + 1]) for i in range(n - half)]))
    gens = BulletproofGens(ctx, 256, table_bits=8)

    def bm_of(bits_):
        out = bytearray((len(bits_) + 7) // 8)
        for i, b in enumerate(bits_):
            out[i // 8] |= b << (i % 8)
        return bytes(out)

    def make():
        bv = BlockVerifier(ctx, gens, batches_in_flight=3)
        bv.set_tx_format(BlockVerifier.TXFORMAT_RECOLLECTED_V1)

        def scenario():
            bm = C.create_string_buffer(b"\xff" * len(want_bm), len(want_bm))
            st = C.create_string_buffer(b"\x00" * n, n)               # (0 = "accepted": the value that must never survive an error)
            rc = lib.zkgpu_tx_verify_batch(bv.h, n, blob, offs, 4, bm, st)
            outs = [(rc, bm.raw, bytes(want_bm))]
            if rc == OK:
                assert st.raw == want_st
            else:
                assert 0 not in st.raw, "a status of 'accepted' beside an error"
            ids = []
            for b_, o_, cnt in ((blob_a, offs_a, half), (blob_b, offs_b, n - half)):
                cid = C.c_uint64(0)
                rc = lib.zkgpu_tx_verify_submit(bv.h, cnt, b_, o_, 2, C.byref(cid))
                ids.append((rc, cid.value, cnt))
            for k, (rc, cid, cnt) in enumerate(ids):
                lo = 0 if k == 0 else half
                wbm = bm_of(list(expected[lo: lo + cnt]))
                if rc != OK:
                    outs.append((rc, bytes(len(wbm)), wbm))
                    continue
                bm2 = C.create_string_buffer(b"\xff" * len(wbm), len(wbm))
                st2 = C.create_string_buffer(b"\x00" * cnt, cnt)
                rc2 = lib.zkgpu_tx_verify_wait(bv.h, cid, bm2, st2)
                outs.append((rc2, bm2.raw, wbm))
                if rc2 == OK:
                    assert st2.raw == want_st[lo: lo + cnt]
                else:
                    assert 0 not in st2.raw
            return outs
        scenario.verifier = bv
        return scenario, bv.close

    try:
        _sweep(ctx, make, _check_outputs("serialized transactions"), "serialized transactions", cold_samples=16)
    finally:
        _disarm(ctx)
        gens.close()
