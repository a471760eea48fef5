"""Oracle: row-shard routing (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Build-defined (the reference is single-device): rows are sharded
by ``owner = id mod G`` with ``local_row = id div G``.
"""
import numpy as np


def bucket_by_owner(ids, world):
    """Stable counting sort of ids by owner.

    Returns (local_rows_sorted[N] int32, counts[G] int64, perm[N] int32) where
    perm[k] is the original position of the k-th element of the sorted list,
    so ``sorted = ids[perm]`` and anything that comes back in sorted order is
    restored with ``out[perm] = back``.
    """
    ids = np.asarray(ids, np.int64)
    owner = ids % world
    perm = np.argsort(owner, kind="stable").astype(np.int32)
    counts = np.bincount(owner, minlength=world).astype(np.int64)
    local = (ids[perm] // world).astype(np.int32)
    return local, counts, perm


def shard_rows(table, world, rank):
    """The rows of ``table`` that rank ``rank`` owns, in local-row order."""
    return table[rank::world]
