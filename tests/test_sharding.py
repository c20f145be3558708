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
