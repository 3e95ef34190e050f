"""Batch dispatch of independent frames over ranks / devices (replicas only, no collective).

Multi-GPU in the reference is one PopSift per device id (popsift.h:158,166-168) with nothing
shared; here one process per GPU takes a contiguous shard of the batch.
"""


def shard_range(n_items, world, rank):
    """Contiguous, balanced [begin, end) shard of n_items for `rank` out of `world`.
    Every item belongs to exactly one rank; shard sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    end = begin + base + (1 if rank < rem else 0)
    return begin, end


def round_robin(n_items, n_slots):
    """frame i -> slot i % n_slots (BASELINE config 4: frame i -> GPU i mod N)."""
    return [i % n_slots for i in range(n_items)]
