"""Sharding a batch of independent verification MSMs over the GPUs of a node -- the torch.distributed
form of what the library does behind its C ABI (zkgpu_shard_cuts + zkgpu_comm_allgather_bitmap,
include/zkgpu.h).

One process per GPU (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests and when
several ranks share one GPU).  Transactions are independent, so there is no data-path collective: each
rank verifies a contiguous block, balanced by the sum of MSM lengths (mixed-arity batches, BASELINE
config 4), and the per-shard accept bitmaps are all-gathered -- a status word + ceil(shard/8) bytes per
rank, latency-bound.  Fail-closed across ranks: a rank whose verifier raised still enters the
collective, flags the error, and every rank then raises with an all-zero result.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple


def partition(offsets: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) row ranges, one per rank, balancing the number of terms (offsets = prefix sums
    of the rows' costs).  Every row lands in exactly one range; ranges may be empty when batch < world.
    Same rule as zkgpu_shard_cuts."""
    batch = len(offsets) - 1
    total = offsets[-1] - offsets[0]
    cuts = [0]
    row = 0
    for r in range(1, world):
        target = offsets[0] + (total * r) // world
        while row < batch and offsets[row + 1] <= target:
            row += 1
        cuts.append(row)
    cuts.append(batch)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def pack_bits(bits: Sequence[int]) -> bytes:
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        if b:
            out[i // 8] |= 1 << (i % 8)
    return bytes(out)


def unpack_bits(bm: bytes, n: int) -> List[int]:
    return [(bm[i // 8] >> (i % 8)) & 1 for i in range(n)]


class ShardError(RuntimeError):
    """A rank of the sharded verification failed; no verdict is returned anywhere."""


def gather_bitmaps(parts: Sequence[Tuple[int, int]], local: bytes, failed: bool, dist, device=None) -> bytes:
    """All-gather of the per-shard bitmaps (+ one status byte per rank) -> bitmap of the whole batch."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    width = 1 + (max((h - l + 7) // 8 for l, h in parts) or 1)
    mine = torch.zeros(width, dtype=torch.uint8)
    mine[0] = 1 if failed else 0
    if local and not failed:
        mine[1: 1 + len(local)] = torch.frombuffer(bytearray(local), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    gathered = torch.zeros(width * world, dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine)
    raw = bytes(gathered.cpu().numpy().tobytes())
    bad = [r for r in range(world) if raw[r * width]]
    if bad:
        raise ShardError("sharded verification failed on rank(s) %s" % bad)
    bits: List[int] = []
    for r, (l, h) in enumerate(parts):
        bits += unpack_bits(raw[r * width + 1:(r + 1) * width], h - l)
    return pack_bits(bits)


def verify_sharded(verify_rows: Callable[[int, int], bytes], offsets: Sequence[int], dist=None, device=None) -> bytes:
    """verify_rows(lo, hi) -> accept bitmap of rows [lo, hi) (bit 0 = row lo).
    Returns the accept bitmap of the WHOLE batch on every rank."""
    batch = len(offsets) - 1
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return verify_rows(0, batch)
    parts = partition(offsets, dist.get_world_size())
    lo, hi = parts[dist.get_rank()]
    local, err = b"", None
    try:
        if hi > lo:
            local = verify_rows(lo, hi)
    except Exception as e:          # still enter the collective: the other ranks must not hang
        err = e
    try:
        return gather_bitmaps(parts, local, err is not None, dist, device)
    except ShardError as e:
        raise (ShardError(str(e)) if err is None else err) from err
