"""
Multi-GPU sharding of the pair x window job.  One process per
GPU; no data-path collective: every pair-window is independent given its two
station windows.  Only the time-stack reduce (day-sharded layout) exchanges
data, through RCCL (``Context.stack_reduce``).
"""
import numpy as np

__all__ = ['all_pairs', 'shard_pairs', 'shard_days', 'window_starts']


def all_pairs(S):
    """Upper triangle i < j, row-major (the reference's pair list order)."""
    return np.array([(i, j) for i in range(S) for j in range(i + 1, S)],
                    dtype=np.int32).reshape(-1, 2)


def shard_pairs(pairs, rank, world):
    """Cyclic split of the pair list: rank r takes pairs r, r+world, ...
    (balanced to within one pair; every pair belongs to exactly one rank)."""
    pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    idx = np.arange(rank, len(pairs), world)
    return np.ascontiguousarray(pairs[idx]), idx


def shard_days(ndays, rank, world):
    """Contiguous split of days (30 days over 8 ranks -> 4,4,4,4,4,4,3,3)."""
    base, extra = divmod(ndays, world)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def window_starts(T, n, hop):
    """Start samples of all whole windows inside a record of T samples."""
    return np.arange(0, T - n + 1, hop, dtype=np.int64)
