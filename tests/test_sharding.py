"""N > 1 path on CPU: world_size-2 gloo processes shard a ragged batch, verify their block
(the oracle stands in for the GPU verifier -- the product has no CPU path) and all-gather the bitmaps."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = 2**252 + 27742317777372353535851937790883648493


def _batch():
    sys.path.insert(0, ROOT)
    from oracle import binding as oracle
    rng = random.Random(5)
    base = oracle.basepoint()
    pts = [oracle.encode(oracle.scalarmult(rng.randrange(1, L), base)) for _ in range(8)]
    sc, pt, offs, want = b"", b"", [0], []
    for i in range(23):
        n = rng.choice([0, 2, 2, 4, 8, 40])
        good = True
        for j in range(n // 2):
            k = rng.randrange(1, L)
            p = pts[(i + j) % 8]
            k2 = (L - k) % L
            if i % 6 == 2 and j == 0:
                k2 = (k2 + 1) % L
                good = False
            sc += k.to_bytes(32, "little") + k2.to_bytes(32, "little")
            pt += p + p
        offs.append(offs[-1] + 2 * (n // 2))
        want.append(int(good))
    return sc, pt, offs, want


def _worker_failing(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from zkvm_amd.sharded import ShardError, verify_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    offs = list(range(0, 41, 2))

    def verify_rows(lo, hi):
        if rank == 1:
            raise RuntimeError("device fault on rank 1")
        return bytes([0xFF]) * ((hi - lo + 7) // 8)

    try:
        verify_sharded(verify_rows, offs, dist)
        q.put((rank, "returned"))
    except ShardError as e:
        q.put((rank, "ShardError"))
    except RuntimeError as e:
        q.put((rank, "own error: " + str(e)))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import binding as oracle
    from zkvm_amd.sharded import verify_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, pt, offs, _ = _batch()

    def verify_rows(lo, hi):
        o = [x - offs[lo] for x in offs[lo:hi + 1]]
        return oracle.verify_batch(sc[32 * offs[lo]: 32 * offs[hi]], pt[32 * offs[lo]: 32 * offs[hi]], o)

    bm = verify_sharded(verify_rows, offs, dist)
    q.put((rank, bm))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_covers_and_balances():
    from zkvm_amd.sharded import partition
    offs = [0, 10, 10, 500, 510, 900, 1400, 1401, 2000]
    for world in (1, 2, 3, 8, 16):
        parts = partition(offs, world)
        assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == len(offs) - 1
        assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
        assert all(lo <= hi for lo, hi in parts)
    two = partition(offs, 2)
    loads = [offs[h] - offs[l] for l, h in two]
    assert abs(loads[0] - loads[1]) <= 600
    assert partition([0], 4) == [(0, 0)] * 4


@pytest.mark.timeout(120)
def test_two_rank_gloo_bitmap_equals_single_process():
    import torch.multiprocessing as mp
    from oracle import binding as oracle
    from zkvm_amd.sharded import unpack_bits
    sc, pt, offs, want = _batch()
    single = oracle.verify_batch(sc, pt, offs)
    assert unpack_bits(single, len(want)) == want
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0] == single and got[1] == single


@pytest.mark.timeout(120)
def test_a_failing_rank_fails_every_rank_and_nobody_hangs():
    """One rank's verifier raises: it still enters the collective with its error flag, every rank gets an error
    instead of a verdict (fail-closed), and the ranks stay in step (the barrier afterwards completes)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_failing, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=100) for _ in range(2))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0] == "ShardError" and got[1].startswith("own error: device fault")


def test_shard_cuts_of_the_library_equal_the_python_partition():
    """zkgpu_shard_cuts (behind the C ABI, no GPU needed) = sharded.partition over the same per-transaction costs
    (terms of the verification MSM of each shape); every transaction in exactly one shard."""
    import random
    from zkvm_amd.native import load_library, shard_cuts
    from zkvm_amd.sharded import partition
    lib = load_library()
    terms = {s: int(lib.zkgpu_cloak_msm_terms(*s)) for s in [(1, 1), (1, 2), (2, 2), (3, 3), (4, 4)]}
    assert terms == {(1, 1): 157, (1, 2): 547, (2, 2): 549, (3, 3): 553, (4, 4): 1071}      # 11 + m + 2k + 2 + 2 pn
    assert lib.zkgpu_cloak_msm_terms(0, 0) == 0 and lib.zkgpu_cloak_msm_terms(65, 1) == 0
    rng = random.Random(3)
    for n in (0, 1, 7, 1000, 8192):
        shapes = [rng.choice(list(terms)) for _ in range(n)]
        offs = [0]
        for s in shapes:
            offs.append(offs[-1] + terms[s])
        for world in (1, 2, 3, 8):
            cuts = shard_cuts(shapes, world)
            assert cuts[0] == 0 and cuts[-1] == n and all(a <= b for a, b in zip(cuts, cuts[1:]))
            assert [(cuts[i], cuts[i + 1]) for i in range(world)] == partition(offs, world)
            if n >= 1000:
                loads = [offs[cuts[i + 1]] - offs[cuts[i]] for i in range(world)]
                assert max(loads) - min(loads) <= 2 * 1071


# ---- the framing of the exchange (zkvm_amd/csrc/comm_frame.hpp through libzkhost.so): any world size on CPU -------------
def _frame_lib():
    import ctypes as C
    from zkvm_amd import build
    build.build()
    lib = C.CDLL(os.path.join(ROOT, "zkvm_amd", "lib", "libzkhost.so"))
    u64p = C.POINTER(C.c_uint64)
    lib.zkhost_comm_slot_bytes.argtypes = [u64p, C.c_int]
    lib.zkhost_comm_slot_bytes.restype = C.c_size_t
    lib.zkhost_comm_pack.argtypes = [C.c_char_p, C.c_size_t, u64p, C.c_int, C.c_char_p, C.c_int]
    lib.zkhost_comm_pack.restype = None
    lib.zkhost_comm_unpack.argtypes = [C.c_char_p, C.c_size_t, u64p, C.c_int, C.c_int, C.c_char_p]
    return lib


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_exchange_framing_status_word_padding_and_cuts(world):
    """What zkgpu_comm_allgather_bitmap puts around the collective, with the collective itself replaced by concatenation:
    ragged shards (empty ones too) land at their cuts bit by bit, the slot is a status word + the widest bitmap padded
    to 8 bytes, ANY rank's non-zero status (its own code, the poison word of a failed copy) gives EVERY rank an error
    and an all-zero bitmap -- its own code on the faulty rank, EREMOTE on the others -- a missing bitmap is a fault that
    travels, and cuts that go backwards are refused identically everywhere."""
    import ctypes as C
    lib = _frame_lib()
    rng = random.Random(100 + world)
    for trial in range(12):
        sizes = [rng.choice([0, 1, 7, 8, 9, 63, 64, 65, 1000, 8192]) for _ in range(world)]
        if trial == 0:
            sizes = [0] * world
        cuts = [0]
        for s in sizes:
            cuts.append(cuts[-1] + s)
        c_cuts = (C.c_uint64 * (world + 1))(*cuts)
        slot = lib.zkhost_comm_slot_bytes(c_cuts, world)
        assert slot == 8 + ((max((s + 7) // 8 for s in sizes) + 7) // 8) * 8
        bits = [rng.getrandbits(1) for _ in range(cuts[-1])]
        local = []
        for r in range(world):
            b = bytearray((sizes[r] + 7) // 8 + 3)                  # trailing garbage beyond the shard must not travel
            for j in range(sizes[r]):
                b[j // 8] |= bits[cuts[r] + j] << (j % 8)
            b[(sizes[r] + 7) // 8:] = b"\xff\xff\xff"
            if sizes[r] % 8:
                b[sizes[r] // 8] |= (0xFF << (sizes[r] % 8)) & 0xFF  # ... nor the spare bits of its last byte
            local.append(bytes(b))
        want = bytearray((cuts[-1] + 7) // 8)
        for i, v in enumerate(bits):
            want[i // 8] |= v << (i % 8)

        def gather(statuses, missing=()):
            buf = b""
            for r in range(world):
                out = C.create_string_buffer(slot)
                lib.zkhost_comm_pack(out, slot, c_cuts, r, None if r in missing else local[r], statuses[r])
                buf += out.raw
            return buf

        def unpack(buf, rank):
            whole = C.create_string_buffer(max(len(want), 1))
            rc = lib.zkhost_comm_unpack(buf, slot, c_cuts, world, rank, whole)
            return rc, whole.raw[: len(want)]

        ok = gather([0] * world)
        for r in range(world):
            assert int.from_bytes(ok[slot * r: slot * r + 4], "little") == 0 and ok[slot * r + 4: slot * r + 8] == bytes(4)
            assert unpack(ok, r) == (0, bytes(want))
        bad_rank = rng.randrange(world)
        for code in (-3, -4, -6):
            st = [0] * world
            st[bad_rank] = code
            buf = gather(st)
            for r in range(world):
                rc, whole = unpack(buf, r)
                assert rc == (code if r == bad_rank else -7) and whole == bytes(len(want))
        # the poison word a rank's send buffer holds between calls (its copy to the device failed): an error everywhere
        poisoned = bytearray(ok)
        poisoned[slot * bad_rank: slot * bad_rank + 4] = (0x80000001).to_bytes(4, "little")
        for r in range(world):
            rc, whole = unpack(bytes(poisoned), r)
            assert rc != 0 and whole == bytes(len(want))
        if sizes[bad_rank]:
            buf = gather([0] * world, missing=(bad_rank,))
            for r in range(world):
                rc, whole = unpack(buf, r)
                assert rc == (-1 if r == bad_rank else -7) and whole == bytes(len(want))
    if world >= 2:
        back = (C.c_uint64 * (world + 1))(*([0, 5, 3] + [9] * (world - 2)))
        assert lib.zkhost_comm_slot_bytes(back, world) == 0
