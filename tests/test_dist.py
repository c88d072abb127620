"""CPU, world_size 2 over gloo: the N > 1 host logic -- cyclic row shards, padded all_gather,
reassembly -- with the oracle standing in for the per-rank compute (there is no GPU here)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle as O
import shard as SH
import synth_genomes as SG


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, seqs = SG.make_set(n, 17, lmin=1500, lmax=3000, fam=3)
    rows = SH.row_shard(n, rank, world)
    shard = np.zeros(SH.shard_len(n, world), dtype=np.int32)
    buf = shard.reshape(SH.shard_rows_max(n, world), n - 1, 3)
    for i, r in enumerate(rows):                 # this rank's rows only
        ref_ids = np.array([r], dtype=np.uint32)
        for j, q in enumerate([q for q in range(n) if q != r]):
            buf[i, j] = O.oracle_pair(seqs[r], seqs[q])
    t = torch.from_numpy(shard)
    gathered = torch.zeros(world * t.numel(), dtype=torch.int32)
    dist.all_gather_into_tensor(gathered, t)
    res = SH.assemble(gathered.numpy(), n, world)
    if rank == 0:
        np.save(out_path, res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 8])
def test_row_shards_gather_world2(tmp_path, n):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    got = np.load(out)
    _, seqs = SG.make_set(n, 17, lmin=1500, lmax=3000, fam=3)
    assert np.array_equal(got, O.oracle_all2all(seqs, None, threads=4))


def test_shard_partition_properties():
    for n in (1, 2, 9, 1000):
        for world in (1, 2, 3, 8):
            rows = np.concatenate([SH.row_shard(n, r, world) for r in range(world)])
            assert sorted(rows.tolist()) == list(range(n))
            assert max(len(SH.row_shard(n, r, world)) for r in range(world)) == SH.shard_rows_max(n, world)
