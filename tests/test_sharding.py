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
