"""Row sharding of the all2all over ranks and reassembly of the gathered results (host-side helpers of
bench.py and the tests; the partition itself is the C-ABI's lzani_partition_rows, the same code the
`lz-ani --gpus n` binary uses through lzani_group_run_rows).

The unit that shards is the reference row (one reference against its query list), the reference's own
work unit (lz_matcher.cpp:196-255).  Dense rows cost the same and are dealt cyclically in the reordered
(length-descending) id order; the ragged rows of a kmer-db filter go through greedy LPT on
cost(row) = sum(Lq) + c * Lr.  The packed genome set is replicated on every GPU; the only exchange is one
gather of the per-pair int32[3] records after the compute.
"""
import numpy as np

import lzani_ctypes as L


def slab_rows(n, step, slab):
    """Reference ids of step `step` of a slab-wise pass over the n rows of an all2all (wraps around)."""
    per_pass = (n + slab - 1) // slab
    s = step % per_pass
    return np.arange(s * slab, min(n, (s + 1) * slab), dtype=np.uint32)


def rank_rows(rows, rank, world, row_cost=None):
    """The rows of `rows` owned by `rank` (lzani_partition_rows: cyclic, or LPT when costs are given)."""
    rows = np.asarray(rows, dtype=np.uint32)
    part = L.partition_rows(len(rows), world, row_cost)
    return rows[part == rank]


def row_shard(n, rank, world):
    """Dense all2all: reference ids owned by `rank` (rank, rank + world, ...)."""
    return rank_rows(np.arange(n, dtype=np.uint32), rank, world)


def shard_rows_max(n_rows, world):
    return (n_rows + world - 1) // world


def shard_len(n, world, n_rows=None):
    """int32 elements of one (padded) shard buffer: rows_max * (n-1) pairs * 3."""
    return shard_rows_max(n if n_rows is None else n_rows, world) * max(n - 1, 0) * 3


def assemble(gathered, n, world, rows=None):
    """gathered: int32[world * shard_len] (rank-major) -> res[n, n, 3], res[r, q] = parse(query=q, ref=r),
    filled for the reference rows `rows` (default: all n)."""
    rows = np.arange(n, dtype=np.uint32) if rows is None else np.asarray(rows, dtype=np.uint32)
    g = np.asarray(gathered, dtype=np.int32).reshape(world, shard_rows_max(len(rows), world), max(n - 1, 0), 3)
    res = np.zeros((n, n, 3), dtype=np.int32)
    for rank in range(world):
        for i, r in enumerate(rank_rows(rows, rank, world)):
            res[r, np.arange(n) != r] = g[rank, i]
    return res
