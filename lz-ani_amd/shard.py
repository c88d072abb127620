"""Row sharding of the all2all over ranks and reassembly of the gathered results.

The unit that shards is the reference row (one reference against its query list), the
reference's own work unit (lz_matcher.cpp:196-255).  Rows are dealt cyclically in the reordered
(length-descending) id order, so every rank gets the same mix of long and short references; the
packed genome set is replicated on every GPU; the only exchange is one all_gather of the
per-pair int32[3] records (equal-size padded shards), after the compute.
"""
import numpy as np


def row_shard(n, rank, world):
    """Reference ids owned by `rank`: rank, rank+world, ..."""
    return np.arange(rank, n, world, dtype=np.uint32)


def shard_rows_max(n, world):
    return (n + world - 1) // world


def shard_len(n, world):
    """int32 elements of one (padded) shard buffer: rows_max * (n-1) pairs * 3."""
    return shard_rows_max(n, world) * max(n - 1, 0) * 3


def assemble(gathered, n, world):
    """gathered: int32[world * shard_len] (rank-major) -> res[n, n, 3], res[r, q] = parse(query=q, ref=r)."""
    g = np.asarray(gathered, dtype=np.int32).reshape(world, shard_rows_max(n, world), max(n - 1, 0), 3)
    res = np.zeros((n, n, 3), dtype=np.int32)
    offdiag = ~np.eye(n, dtype=bool)
    for rank in range(world):
        rows = row_shard(n, rank, world)
        for i, r in enumerate(rows):
            res[r, offdiag[r]] = g[rank, i]
    return res
