"""CPU, world_size 2 over gloo: the N > 1 host logic -- the C-ABI's row partition (lzani_partition_rows /
lzani_row_costs: the same code lzani_group_run_rows and bench.py use), padded all-gather, ragged gather,
reassembly -- with the oracle standing in for the per-rank compute (there is no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import lzani_ctypes as L
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
            assert SH.row_shard(n, 0, world).tolist() == list(range(0, n, world))        # cyclic
    # slabs of a pass cover every row once and wrap around
    n, slab = 1030, 500
    seen = np.concatenate([SH.slab_rows(n, s, slab) for s in range(3)])
    assert sorted(seen.tolist()) == list(range(n))
    assert np.array_equal(SH.slab_rows(n, 3, slab), SH.slab_rows(n, 0, slab))


def heavy_tailed_csr(n_target, seed):
    """A symmetric sparse pair list as a kmer-db prefilter leaves it: families of heavy-tailed size (40 % singletons,
    40 % 2-10, 15 % 11-100, 5 % 101-400), all same-family pairs plus one random cross pair per genome."""
    st = SG.Stream(seed)
    fam_of = []
    while len(fam_of) < n_target:
        u = st.one()
        sz = 1 if u < 0.4 else st.randint(2, 10) if u < 0.8 else st.randint(11, 100) if u < 0.95 else st.randint(101, 400)
        fam_of += [len(set(fam_of))] * min(sz, n_target - len(fam_of))
    fam_of = np.array(fam_of)
    n = len(fam_of)
    rows = [[] for _ in range(n)]
    start = 0
    for f in range(fam_of.max() + 1):
        m = int((fam_of == f).sum())
        for a in range(start, start + m):
            rows[a] += [b for b in range(start, start + m) if b != a]
        start += m
    for a in range(n):
        b = st.randint(0, n - 1)
        if b != a and b not in rows[a]:
            rows[a].append(b)
            rows[b].append(a)
    lens = np.array([st.randint(30000, 50000) for _ in range(n)], dtype=np.uint32)
    ref_ids = np.arange(n, dtype=np.uint32)
    row_off = np.zeros(n + 1, dtype=np.uint64)
    row_off[1:] = np.cumsum([len(r) for r in rows])
    q = np.array([x for r in rows for x in r], dtype=np.uint32)
    return ref_ids, row_off, q, lens


def test_lpt_partition_balances_heavy_tailed_rows():
    """BASELINE configs[4] shape: the cost of a filtered row spans three orders of magnitude; greedy LPT on
    cost(row) = sum(Lq) + c*Lr keeps the heaviest shard within 10 % of the mean at 2, 4 and 8 GPUs, while
    dealing the same rows cyclically does not."""
    ref_ids, row_off, q, lens = heavy_tailed_csr(6000, 21)
    cost = L.row_costs(ref_ids, row_off, q, lens)
    # the cost is what the header says it is
    k = 4321
    assert int(cost[k]) == int(lens[q[int(row_off[k]):int(row_off[k + 1])]].astype(np.int64).sum()) + 6 * int(lens[k])
    sizes = np.diff(row_off.astype(np.int64))
    assert sizes.max() > 100 * max(1, sizes.min()) and (sizes <= 3).mean() > 0.01    # really heavy-tailed
    for world in (2, 4, 8):
        part = L.partition_rows(len(ref_ids), world, cost)
        assert part.min() == 0 and part.max() == world - 1
        load = np.bincount(part, weights=cost.astype(np.float64), minlength=world)
        assert load.max() / load.mean() <= 1.1, (world, load.max() / load.mean())
        # the algorithm is the stated one: heaviest row first onto the least loaded shard (ties: lowest shard)
        want = np.zeros(len(cost), dtype=np.uint32)
        acc = [0] * world
        for r in np.argsort(-cost.astype(np.int64), kind="stable"):
            p = min(range(world), key=lambda x: (acc[x], x))
            want[r] = p
            acc[p] += int(cost[r])
        assert np.array_equal(part, want)
    cyc = L.partition_rows(len(ref_ids), 8, None)
    assert np.array_equal(cyc, np.arange(len(ref_ids)) % 8)
    with pytest.raises(L.LzaniError):
        L.partition_rows(3, 0, None)


def _worker_csr(rank, world, port, out_path):
    """Filtered rows: every rank takes its LPT shard, computes it (oracle), and the ragged shards are gathered
    to rank 0 in rank order -- the data movement of lzani_comm_gatherv / lzani_group_run_rows -- and put back
    into the caller's CSR order."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _, seqs = SG.make_set(40, 23, lmin=1200, lmax=2500, fam=8)
    ref_ids, row_off, q, _ = heavy_tailed_csr(40, 5)
    lens = np.array([len(s) for s in seqs], dtype=np.uint32)
    part = L.partition_rows(len(ref_ids), world, L.row_costs(ref_ids, row_off, q, lens))
    counts = [int(np.diff(row_off.astype(np.int64))[part == p].sum()) for p in range(world)]
    mine = np.nonzero(part == rank)[0]
    shard = np.zeros((max(counts), 3), dtype=np.int32)                        # padded to the largest shard for gloo
    at = 0
    for k in mine:
        for e in range(int(row_off[k]), int(row_off[k + 1])):
            shard[at] = O.oracle_pair(seqs[ref_ids[k]], seqs[q[e]])
            at += 1
    assert at == counts[rank]
    bufs = [torch.zeros(shard.shape, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(bufs, torch.from_numpy(shard))
    if rank == 0:
        out = np.zeros((len(q), 3), dtype=np.int32)
        for p in range(world):
            got, at = bufs[p].numpy(), 0
            for k in np.nonzero(part == p)[0]:
                cnt = int(row_off[k + 1] - row_off[k])
                out[int(row_off[k]):int(row_off[k + 1])] = got[at:at + cnt]
                at += cnt
        np.save(out_path, out)
    dist.barrier()
    dist.destroy_process_group()


def test_filtered_rows_lpt_gather_world2(tmp_path):
    out = str(tmp_path / "csr.npy")
    mp.spawn(_worker_csr, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    _, seqs = SG.make_set(40, 23, lmin=1200, lmax=2500, fam=8)
    ref_ids, row_off, q, _ = heavy_tailed_csr(40, 5)
    full = O.oracle_all2all(seqs, None, threads=4)
    e = 0
    for k, r in enumerate(ref_ids):
        for x in q[int(row_off[k]):int(row_off[k + 1])]:
            assert tuple(got[e]) == tuple(full[r, x]), (k, r, x)
            e += 1
    assert e == len(q)


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu():
    """bench.py's N > 1 logic on hardware: two ranks under torch.distributed.run share GPU 0 (RCCL refuses that, so the
    all-gather goes through gloo), slabs dealt cyclically by lzani_partition_rows, every rank's HIP path, the gathered
    slab checked against the oracle on rows of both ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--genomes", "300", "--seed", "9", "--slab", "50", "--lmin", "3000", "--lmax", "5000", "--collective", "gloo", "--device", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.split("\n") if ln.startswith("{")][0]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["parity_on_last_slab"] == "bit-exact"
    # a step = 50 rows per rank: 100 rows in all, dealt cyclically; the per-rank split of the timed region is there for both ranks
    assert d["config"]["pairs_timed"] == 3 * 100 * 299 and d["config"]["pairs_timed_rank0"] == 3 * 50 * 299
    assert d["config"]["slab_rows"] == 100 and d["config"]["slab_rows_per_rank"] == 50
    assert all(len(d["per_rank_ms_per_step"][k]["by_rank"]) == 2 for k in ("compute_ms", "kernel_ms", "fixed_ms", "gather_ms"))


def test_group_gather_plan_against_python_statement():
    """The shard bookkeeping of lzani_group_run_rows (lzani_plan_gather, a pure host function of the C-ABI) against
    a Python statement of the same rule, for ragged rows with empty rows, empty shards and more shards than rows:
    shards are contiguous in the gathered buffer in shard order, rows keep their order inside a shard, and the
    scatter table moves every row's results to its place in the caller's CSR order (simulated here on arrays)."""
    st = SG.Stream(91)
    cases = []
    for n_rows, n_parts in ((0, 3), (1, 4), (5, 8), (37, 2), (200, 8), (64, 3)):
        sizes = [0 if st.one() < 0.15 else st.randint(1, 40) for _ in range(n_rows)]
        part = [st.randint(0, n_parts - 1) for _ in range(n_rows)]
        cases.append((sizes, part, n_parts))
    cases.append(([3, 0, 7, 2], [2, 2, 2, 2], 4))               # every row on one shard, three shards empty
    cases.append(([5] * 12, [k % 8 for k in range(12)], 8))     # the cyclic deal of dense rows
    for sizes, part, n_parts in cases:
        n_rows = len(sizes)
        row_off = np.zeros(n_rows + 1, dtype=np.uint64)
        row_off[1:] = np.cumsum(sizes)
        base, src, dst, cnt, row = L.plan_gather(row_off, np.array(part, dtype=np.uint32), n_parts)
        # the rule, restated
        want_base = [0]
        for d in range(n_parts):
            want_base.append(want_base[-1] + sum(s for s, p in zip(sizes, part) if p == d))
        assert base.tolist() == want_base
        j = 0
        for d in range(n_parts):
            at = want_base[d]
            for k in range(n_rows):
                if part[k] != d:
                    continue
                assert (int(src[j]), int(dst[j]), int(cnt[j]), int(row[j])) == (at, int(row_off[k]), sizes[k], k), (d, k)
                at += sizes[k]
                j += 1
        assert j == n_rows
        # what the devices and the scatter kernel do with it: shard d computes its rows in order into its part of the
        # gathered buffer; the table then restores the CSR order
        n_pairs = int(row_off[-1])
        csr = np.arange(n_pairs, dtype=np.int64) * 7 + 1            # a distinct value per pair, in the caller's order
        gathered = np.full(n_pairs, -1, dtype=np.int64)
        for d in range(n_parts):
            at = want_base[d]
            for k in range(n_rows):
                if part[k] == d:
                    gathered[at:at + sizes[k]] = csr[int(row_off[k]):int(row_off[k + 1])]
                    at += sizes[k]
        final = np.full(n_pairs, -2, dtype=np.int64)
        for j in range(n_rows):
            final[int(dst[j]):int(dst[j]) + int(cnt[j])] = gathered[int(src[j]):int(src[j]) + int(cnt[j])]
        assert np.array_equal(final, csr)
    with pytest.raises(L.LzaniError):
        L.plan_gather(np.array([0, 2], np.uint64), np.array([5], np.uint32), 2)        # a shard number out of range
