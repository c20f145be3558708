"""Sharding a batch of independent verification MSMs over the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  Transactions are independent, so there is no
data-path collective: each rank verifies a contiguous block, balanced by the sum
of MSM lengths (mixed-arity batches, BASELINE config 4), and the per-shard accept
bitmaps are all-gathered -- ceil(shard/8) bytes per rank, latency-bound.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple


def partition(offsets: Sequence[int], world: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) row ranges, one per rank, balancing the number of terms.
    Every row lands in exactly one range; ranges may be empty when batch < world."""
    batch = len(offsets) - 1
    total = offsets[-1] - offsets[0]
    cuts = [0]
    row = 0
    for r in range(1, world):
        target = offsets[0] + (total * r) // world
        while row < batch and offsets[row + 1] - 0 <= target:
            row += 1
        # keep cuts monotone and leave at least the remaining ranks something when possible
        row = max(row, cuts[-1])
        cuts.append(min(row, batch))
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


def verify_sharded(verify_rows: Callable[[int, int], bytes], offsets: Sequence[int], dist=None, device=None) -> bytes:
    """verify_rows(lo, hi) -> accept bitmap of rows [lo, hi) (bit 0 = row lo).
    Returns the accept bitmap of the WHOLE batch on every rank."""
    import torch
    batch = len(offsets) - 1
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return verify_rows(0, batch)
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = partition(offsets, world)
    lo, hi = parts[rank]
    local = verify_rows(lo, hi) if hi > lo else b""
    width = max((h - l + 7) // 8 for l, h in parts) or 1
    mine = torch.zeros(width, dtype=torch.uint8)
    if local:
        mine[: len(local)] = torch.frombuffer(bytearray(local), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    gathered = torch.zeros(width * world, dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine)
    raw = bytes(gathered.cpu().numpy().tobytes())
    bits: List[int] = []
    for r, (l, h) in enumerate(parts):
        bits += unpack_bits(raw[r * width:(r + 1) * width], h - l)
    return pack_bits(bits)
