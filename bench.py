#!/usr/bin/env python3
"""bench.py -- all2all ANI hot path on MI355X: directed genome pairs per second.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1: under torch.distributed.run)
prints ONE JSON line on rank 0.

Workload = the configuration BASELINE.json's metric is quoted on (configs[2]):
10,000 synthetic viral genomes of ~40 kbp (tools/synth_genomes.py, seed 2: families of 10, 1-15 %
divergence), default LZ parameters, dense all2all = 99,990,000 directed pairs -- at EVERY N; a step holds
500 rows per rank ("weak" scaling per step: a GPU's batch is what it is in the product whatever N; the job is
covered in 20 / N steps).

A step = one SLAB of the all2all: 500 consecutive reference rows PER RANK (in the reference's length-descending
order) against all other genomes = 4,999,500 directed pairs per rank, i.e. one pass of the hot path over one batch:
per-reference index build + candidate stage + pair kernel on every rank's share of the slab (rows dealt cyclically over the
ranks by the C-ABI's lzani_partition_rows) + one RCCL all-gather of the per-pair int32[3] records (N > 1:
torch.distributed's nccl backend by default, `--collective lzani` = lzani_comm_allgather inside the engine
library, which no pool has yet let run on more than one GPU).  `--steps 20` at N = 1 is exactly one pass over the 10k x 10k matrix; the slabs wrap
around.  Genomes are resident in HBM before the timed region; results stay in HBM.

`roofline` is for the pair kernel (k_pairs): algorithmic bytes B_pair summed over the pairs
of rank 0's timed launches, divided by the summed launch durations measured with HIP events on the
engine's stream.  `roofline.bytes_definition` names the byte count; `achieved_incl_candidate_stage` divides the same bytes by
pair kernel + candidate stage (k_pm_build / k_pm_cand read the k-mer-word stream that B_pair prices, since round 3).

`--workload related` is the shape a kmer-db prefilter leaves (BASELINE configs[4]): genomes in families of `--fam`, every row
holds the same-family queries only (filtered CSR rows, all pairs related at <= `--dmax` divergence); a step = `--slab` rows.  `cpu_baseline` times the reference's own CParser (oracle/_ref, kind "reference"; the C
restatement as kind "port" if that is absent) on a strided sample of the same workload, rank 0, N = 1 only.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lz-ani_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import lzani_ctypes as L  # noqa: E402
import shard as SH  # noqa: E402
import synth_genomes as SG  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)


def pair_bytes(lens, params):
    """Per genome: its bytes as a query (ceil(Lq/4) + 4(Lq+mrd-mal+1) + 12) and as a reference (ceil((2Lr+3mrd)/4))."""
    lens = lens.astype(np.int64)
    mrd, mal = params["mrd"], params["mal"]
    q_bytes = (lens + 3) // 4 + 4 * np.maximum(lens + mrd - mal + 1, 0) + 12
    r_bytes = (2 * lens + 3 * mrd + 3) // 4
    return q_bytes, r_bytes


def algorithmic_bytes(lens, ref_ids, params):
    """Sum over the dense rows `ref_ids` of B_pair = ceil(Lq/4) + ceil((2Lr+3mrd)/4) + 4(Lq+mrd-mal+1) + 12."""
    q_bytes, r_bytes = pair_bytes(lens, params)
    n = len(lens)
    ref_ids = np.asarray(ref_ids, dtype=np.int64)
    return int(((n - 1) * r_bytes[ref_ids] + (int(q_bytes.sum()) - q_bytes[ref_ids])).sum())


def algorithmic_bytes_csr(lens, ref_ids, row_off, query_ids, params):
    """The same sum over filtered rows (ref_ids, row_off, query_ids)."""
    q_bytes, r_bytes = pair_bytes(lens, params)
    cnt = np.diff(np.asarray(row_off, dtype=np.int64))
    return int((r_bytes[np.asarray(ref_ids, dtype=np.int64)] * cnt).sum() + q_bytes[np.asarray(query_ids, dtype=np.int64)].sum())


def pmc_traffic(n, seed, slab, world):
    """HBM bytes per k_pairs launch from the committed rocprofv3 PMC passes (profiles/latest_pmc.json),
    quoted only when they were collected on this very workload; PMC cannot be read from inside the run."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_pmc.json")) as f:
            rec = json.load(f)
        w = rec["workload"]
        if world == 1 and w["genomes"] == n and w["seed"] == seed and w.get("slab_rows") == slab:
            return rec["traffic_bytes_fetch_doubled"]
    except Exception:
        pass
    return None


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(math.ceil(int(quota) / int(period)))))
    except Exception:
        pass
    return cores


def cpu_baseline(seqs, params, sample_ids):
    """Reference CParser (or the C port) on the dense all2all of the sampled genomes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    cores = host_cores()
    sub = [seqs[i] for i in sample_ids]
    npairs = len(sub) * (len(sub) - 1)
    kind = "reference" if O.lib_ref() is not None else "port"
    run = O.ref_all2all if kind == "reference" else O.oracle_all2all
    t = time.perf_counter()
    res = run(sub, params, threads=cores)
    dt = time.perf_counter() - t
    what = ("the reference's own CParser (parser.cpp + utils.cpp compiled unmodified) behind oracle/ref_driver.cpp = the reference's "
            "'LZ matching' stage without its FASTA ingest and TSV output (the full lz-ani binary builds only with a stand-in for "
            "its un-vendored zlib-ng header, which this repo does not write; the matching stage alone is the like-for-like baseline)"
            if kind == "reference" else "oracle/lzani_oracle.c, the C restatement")
    return dict(value=npairs / dt, unit="genome-pairs/s", cores=cores, kind=kind,
                sample=f"dense all2all of every {max(1, len(seqs) // len(sub))}-th genome of the workload in the reference's length-descending "
                       f"order ({len(sub)} genomes, {npairs} pairs, {dt:.2f} s wall = {dt * cores:.0f} s of CPU work, {cores} threads "
                       f"self-scheduling over reference rows as in do_matching); code = {what}"), res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="dense", choices=("dense", "related"),
                    help="dense: the metric's all2all (BASELINE configs[2]); related: filtered rows of same-family pairs (configs[4] shape)")
    ap.add_argument("--genomes", type=int, default=10000, help="genomes of the set (BASELINE configs[2]: 10,000)")
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--slab", type=int, default=0, help="reference rows per step (per RANK with --scaling weak); 0 = 500 (dense) / every row (related)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="N > 1: weak = --slab rows per rank and step (a rank's batch is what it is in the product whatever N); "
                         "strong = --slab rows per step shared out over the ranks (the per-step fixed cost -- candidate stage, index build -- "
                         "does not shrink with a rank's rows)")
    ap.add_argument("--fam", type=int, default=50, help="--workload related: family size")
    ap.add_argument("--dmax", type=float, default=0.15, help="--workload related: largest per-base divergence from the ancestor")
    ap.add_argument("--cpu-sample", type=int, default=192, help="genomes in the CPU baseline sample (0 = skip)")
    ap.add_argument("--lmin", type=int, default=36000, help="ancestor length range of the synthetic set")
    ap.add_argument("--lmax", type=int, default=44000)
    ap.add_argument("--params", default="", help="LZ parameter overrides, e.g. mal=15,msl=9,reg=60 (BASELINE configs[3])")
    ap.add_argument("--collective", default="torch", choices=("lzani", "torch", "gloo"),
                    help="torch: torch.distributed nccl (= RCCL) all-gather, the default until the library's own communicator has run on "
                         "more than one GPU; lzani: RCCL all-gather inside the engine library (lzani_comm_allgather); "
                         "gloo: host-side all_gather -- only for rehearsing the N > 1 logic with several ranks on ONE GPU (--device 0), "
                         "where RCCL refuses to run")
    ap.add_argument("--device", type=int, default=-1, help="force the HIP device ordinal")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle check of the last slab (diagnostic builds that skip work on purpose)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    related = args.workload == "related"
    if related and world > 1:
        raise SystemExit("--workload related is a one-GPU diagnostic workload (the metric's own configuration is --workload dense)")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = local_rank if args.device < 0 else args.device
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.collective == "torch":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend="gloo")          # rendezvous, unique id, barriers; the data path is RCCL in the library

    # rows of a step: weak = `--slab` per rank, so that a rank's share of a step -- one index build, one candidate stage (whose
    # cost does not shrink with the rows), one pair-kernel launch -- is the same at every N, as in the product, where a GPU works
    # through its share of the matrix in batches of hundreds of rows whatever the number of GPUs (the job stays the
    # 10,000 x 10,000 all2all, covered in 20 / N steps); strong = `--slab` rows per step in all
    n = args.genomes
    slab_arg = args.slab if args.slab > 0 else (n if related else 500)
    slab_rank = max(1, min(slab_arg, n))
    slab = max(1, min(slab_rank * world, n)) if args.scaling == "weak" else slab_rank
    over = {k: int(v) for k, v in (kv.split("=") for kv in args.params.split(",") if kv)}
    if related:
        names, seqs = SG.make_set_cached(n, args.seed, lmin=args.lmin, lmax=args.lmax, fam=args.fam, dmax=args.dmax)
    else:
        names, seqs = SG.make_set_cached(n, args.seed, lmin=args.lmin, lmax=args.lmax)
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    # reference order: length-descending, then name (CSeqReservoir::reorder_items, seq_reservoir.cpp:215-251)
    order = sorted(range(n), key=lambda i: (-int(lens[i]), names[i]))
    seqs = [seqs[i] for i in order]
    lens = lens[order]
    fam_of = np.array(order, dtype=np.int64) // max(1, args.fam)      # (related) family of the genome at a reordered id
    members = {}
    if related:
        for g, f in enumerate(fam_of.tolist()):
            members.setdefault(f, []).append(g)

    eng = L.Engine(over or None, device=dev)
    params = eng.params
    t_set = time.perf_counter()
    eng.set_genomes(seqs)                      # untimed: genomes resident in HBM before the timed region
    set_genomes_ms = (time.perf_counter() - t_set) * 1e3
    if world > 1 and args.collective == "lzani":
        box = [L.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        eng.comm_init(world, rank, box[0])

    csr_cache = {}

    def csr_of(rows):
        """Filtered rows of the related workload: every reference against the other members of its family (the row tables are
        the caller's input -- what a kmer-db filter hands over -- and are made once per slab, outside the timed region)."""
        key = (int(rows[0]), len(rows)) if len(rows) else (0, 0)
        if key not in csr_cache:
            qs = [[q for q in members[int(fam_of[r])] if q != int(r)] for r in rows]
            row_off = np.zeros(len(rows) + 1, dtype=np.uint64)
            row_off[1:] = np.cumsum([len(x) for x in qs])
            csr_cache[key] = (np.asarray(rows, dtype=np.uint32), row_off, np.array([x for r in qs for x in r], dtype=np.uint32))
        return csr_cache[key]

    rows_max = SH.shard_rows_max(slab, world)
    per_rank = rows_max * ((args.fam - 1) if related else (n - 1))    # padded shard, in results
    shard = torch.zeros(max(1, per_rank) * 3, dtype=torch.int32, device="cuda")
    gathered = torch.zeros(world * per_rank * 3, dtype=torch.int32, device="cuda") if world > 1 else None
    host_parts = [torch.zeros(per_rank * 3, dtype=torch.int32) for _ in range(world)] if args.collective == "gloo" else None

    def step(s, acc=None):
        rows = SH.slab_rows(n, s, slab)
        mine = SH.rank_rows(rows, rank, world)
        if related:
            ref_ids, row_off, q_ids = csr_of(mine)
        else:
            (ref_ids, row_off), q_ids = L.dense_rows(n, mine), None
        t_a = time.perf_counter()
        eng.run_rows_device(ref_ids, row_off, q_ids, shard.data_ptr())      # (blocking: returns when the shard is complete)
        t_b = time.perf_counter()
        if world > 1:
            if args.collective == "lzani":
                eng.comm_allgather(shard.data_ptr(), gathered.data_ptr(), per_rank)
            elif args.collective == "torch":
                dist.all_gather_into_tensor(gathered, shard)
            else:
                dist.all_gather(host_parts, shard.cpu())
                gathered.copy_(torch.cat(host_parts))
            if acc is not None:
                torch.cuda.synchronize()                                    # (the gather on this rank, for the per-rank split below)
        if acc is not None:
            tm = eng.timing()
            acc["compute_ms"] += (t_b - t_a) * 1e3
            acc["gather_ms"] += (time.perf_counter() - t_b) * 1e3
            acc["kernel_ms"] += tm["pairs_ms"]
            acc["fixed_ms"] += tm["index_ms"] + tm["cand_ms"] + tm["kmers_ms"]   # not sharded by rows: k-mer words (first run only), index build, candidate stage
        return rows, mine, (ref_ids, row_off, q_ids)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if related:                                               # (the row tables of every slab the run will touch)
        for s in range(args.warmup + args.steps):
            csr_of(SH.rank_rows(SH.slab_rows(n, s, slab), rank, world))
    for s in range(args.warmup):
        step(s)
    fence()
    kernel_ms, index_ms, cand_ms, launches, abytes, total_pairs, my_pairs = 0.0, 0.0, 0.0, 0, 0, 0, 0
    split = {"compute_ms": 0.0, "kernel_ms": 0.0, "fixed_ms": 0.0, "gather_ms": 0.0}
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        rows, mine, csr = step(s, split)
        tm = eng.timing()
        kernel_ms += tm["pairs_ms"]
        index_ms += tm["index_ms"]
        cand_ms += tm["cand_ms"]
        launches += tm["pair_launches"]
        if related:
            total_pairs += int(csr[1][-1])
            my_pairs += int(csr[1][-1])
            abytes += algorithmic_bytes_csr(lens, *csr, params)
        else:
            total_pairs += len(rows) * (n - 1)
            my_pairs += len(mine) * (n - 1)
            if rank == 0:
                abytes += algorithmic_bytes(lens, mine, params)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.collective == "torch" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # per-rank split of the timed region (so that a scaling curve can be read the moment a node exists): host wall of the
    # blocking compute call, device time of the pair kernel and of the stages that do not shrink with the rank's share
    # of rows, host wall of the gather
    splits = [split]
    if world > 1:
        splits = [None] * world
        dist.all_gather_object(splits, split)

    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_pairs / dt
        launches = max(1, launches)
        avg_launch_ms = kernel_ms / launches
        achieved = abytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        achieved_cand = abytes / ((kernel_ms + cand_ms) * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        lay = eng.layout()
        kernel_name = "k_pairs_blk" if lay["block_launches"] else "k_pairs (candidate bitmaps)" if lay["bitmap_launches"] else "k_pairs"
        coll_name = {"lzani": "lzani_comm_allgather", "torch": "torch.distributed nccl",
                     "gloo": "REHEARSAL: gloo through host memory, ranks may share a GPU"}[args.collective]
        traffic = None if related else pmc_traffic(n, args.seed, slab_rank, world)
        pairs_per_step = total_pairs // max(1, args.steps) if related else slab * (n - 1)
        if related:
            wl = (f"{n} synthetic genomes of {args.lmin}-{args.lmax} bp in families of {args.fam} (divergence <= {args.dmax}, seed {args.seed}), "
                  f"filtered rows: every genome against the other members of its family ({n * (args.fam - 1)} directed pairs per pass, all related), "
                  f"{'default LZ params' if not over else 'LZ params ' + args.params}; one step = {slab} rows")
            metric = "genome-pairs/sec, related pairs only (the shape a kmer-db prefilter leaves: BASELINE configs[4]), 1 GPU"
        else:
            wl = (f"{n} synthetic genomes of {args.lmin}-{args.lmax} bp (families of 10, 1-15% divergence, seed {args.seed}), "
                  f"dense all2all ({n * (n - 1)} directed pairs per pass), "
                  f"{'default LZ params' if not over else 'LZ params ' + args.params}; one step = a slab of {slab_rank} reference rows "
                  f"{'per rank' if args.scaling == 'weak' else 'shared by the ranks'} ({slab} rows in all) x all other genomes = "
                  f"{slab * (n - 1)} directed pairs, {(n + slab - 1) // slab} steps per pass")
            metric = "genome-pairs/sec + achieved HBM GB/s, 10k×40kbp all2all at 1/2/4/8 GPUs"
        fixed_ms = max(sp["fixed_ms"] for sp in splits) / args.steps
        out = {
            "metric": metric,
            "value": value, "unit": "genome-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32 (2-bit packed symbols, 64-bit lane masks; f64 only in the anchor/seed arbitration)",
            "data": "synthetic",
            "config": {"workload": wl,
                       "genomes": n, "seed": args.seed, "pairs_per_pass": n * (args.fam - 1) if related else n * (n - 1),
                       "slab_rows": slab, "slab_rows_per_rank": slab_rank if args.scaling == "weak" else rows_max,
                       "scaling_mode": (f"{args.scaling}: " + ("every rank runs --slab rows per step" if args.scaling == "weak"
                                                              else "--slab rows per step are shared out over the ranks")
                                        + f"; per-step cost that does not shrink with a rank's rows (index build + candidate stage): "
                                          f"{fixed_ms:.1f} ms of {ms_per_step:.1f} ms"),
                       "pairs_per_step": pairs_per_step,
                       "pairs_timed": total_pairs, "pairs_timed_rank0": my_pairs,
                       "sharding": (f"rows of a slab dealt cyclically over {world} ranks (lzani_partition_rows), genomes replicated, one RCCL "
                                    f"all-gather of int32[3] per pair per step ({coll_name})")
                                   if world > 1 else "single GPU, all rows",
                       "params": params,
                       "index_form": {"dir_bits": lay["dir_bits"], "tag_words": lay["tag_words"], "bucket_table": lay["bucket_table"],
                                      "n_free": lay["n_free"], "slots": lay["slots"], "batches_per_step": lay["batches_last_run"],
                                      "block_kernel_with_lds_filter": int(lay["block_launches"] > 0),
                                      "candidate_bitmaps_from_presence_matrix": int(lay["bitmap_launches"] > 0)}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_name, "avg_launch_ms": avg_launch_ms, "launches": launches,
                         "algorithmic_bytes_per_launch": abytes / launches,
                         "bytes_definition": "B_pair of SURVEY 8(d) = ceil(Lq/4) + ceil((2Lr+3mrd)/4) + 4(Lq+mrd-mal+1) + 12 per directed pair: "
                                             "ALGORITHMIC bytes (a work-normalised rate, not a bandwidth utilisation); `achieved` divides by the "
                                             "pair kernel's launch time alone, `achieved_incl_candidate_stage` by pair kernel + candidate stage, "
                                             "which since round 3 reads the 4(Lq+mrd-mal+1) k-mer-word stream in the pair kernel's place",
                         "achieved_incl_candidate_stage": achieved_cand, "frac_incl_candidate_stage": achieved_cand / HBM_PEAK_GBS,
                         "fabric_TBps_from_counters": (traffic / (avg_launch_ms * 1e-3) / 1e12) if traffic and avg_launch_ms > 0 else None,
                         "fabric_note": "traffic = TCC-side fetch bytes of the committed PMC passes (Infinity-Cache hits are inside it: the "
                                        "HBM share is unknown); fabric_TBps = traffic / avg_launch_ms",
                         "index_build_ms_per_step": index_ms / args.steps, "candidate_stage_ms_per_step": cand_ms / args.steps},
            "set_genomes_ms": set_genomes_ms,
            "per_rank_ms_per_step": {k: {"by_rank": [round(sp[k] / args.steps, 3) for sp in splits],
                                         "max": round(max(sp[k] for sp in splits) / args.steps, 3),
                                         "mean": round(sum(sp[k] for sp in splits) / len(splits) / args.steps, 3)}
                                     for k in ("compute_ms", "kernel_ms", "fixed_ms", "gather_ms")},
        }
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        n_check = int(max(4, min(64, 40e6 / max(1.0, float(lens.mean())))))       # oracle pairs checked: ~40 M symbols of CPU work
        if world == 1:
            # PCIe-inclusive rate: the first slab once more through lzani_run_rows with HOST result buffers (upload of the row
            # tables, every stage, results over PCIe into pageable memory); never `value`
            r0 = SH.slab_rows(n, 0, slab)
            a0 = csr_of(r0) if related else (*L.dense_rows(n, r0), None)
            t_p = time.perf_counter()
            eng.run_rows(*a0)
            out["pcie_inclusive_value"] = int(a0[1][-1]) / (time.perf_counter() - t_p)
        if world == 1 and args.cpu_sample > 1 and related:
            # the reference's CParser on the rows of the first families: bounded CPU work on the same kind of pairs
            import oracle as O
            m = min(n, args.cpu_sample * 8)
            rr, ro, rq = csr_of(np.arange(m, dtype=np.uint32))
            cores = host_cores()
            kind = "reference" if O.lib_ref() is not None else "port"
            t = time.perf_counter()
            if kind == "reference":
                cpu_res = O.ref_rows(seqs, rr, ro, rq, params, cores)
            else:
                cpu_res = np.array([O.oracle_pair(seqs[int(r)], seqs[int(q)], params) for k, r in enumerate(rr) for q in rq[int(ro[k]):int(ro[k + 1])]], dtype=np.int32)
            dtc = time.perf_counter() - t
            out["cpu_baseline"] = dict(value=int(ro[-1]) / dtc, unit="genome-pairs/s", cores=cores, kind=kind,
                                       sample=f"the filtered rows of the first {m} genomes of the workload ({int(ro[-1])} related pairs, {dtc:.2f} s wall, "
                                              f"{cores} threads); code = the reference's CParser behind oracle/ref_driver.cpp" if kind == "reference"
                                              else f"the filtered rows of the first {m} genomes ({int(ro[-1])} pairs, {dtc:.2f} s, 1 thread); oracle/lzani_oracle.c")
            got = eng.run_rows(rr, ro, rq)
            out["cpu_baseline"]["parity_on_sample"] = "bit-exact" if np.array_equal(got, np.asarray(cpu_res).reshape(-1, 3)) else "MISMATCH"
            out["cpu_baseline"]["parity_on_sample_kernel"] = kernel_name + " = the timed kernel"
        elif world == 1 and args.cpu_sample > 1:
            m = min(args.cpu_sample, n)
            sample_ids = np.arange(0, n, max(1, n // m), dtype=np.uint32)[:m]
            cb, cpu_res = cpu_baseline(seqs, params, sample_ids)
            out["cpu_baseline"] = cb
            # free parity evidence: the same sampled pairs through the HIP path, untimed, as the dense all2all of the sample in a
            # context of its own -- i.e. through the very kernel the timed region launches (dense rows)
            eng2 = L.Engine(over or None, device=dev)
            eng2.set_genomes([seqs[i] for i in sample_ids])
            got = eng2.all2all()
            lay2 = eng2.layout()
            eng2.close()
            k2 = "k_pairs_blk" if lay2["block_launches"] else "k_pairs (candidate bitmaps)" if lay2["bitmap_launches"] else "k_pairs"
            out["cpu_baseline"]["parity_on_sample"] = "bit-exact" if np.array_equal(got, cpu_res) else "MISMATCH"
            out["cpu_baseline"]["parity_on_sample_kernel"] = k2 + (" = the timed kernel" if k2 == kernel_name else " (NOT the timed kernel)")
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_check and related:
            # pairs of the last timed step, straight from the bench's own result buffer, against the reference's CParser
            import oracle as O
            rr, ro, rq = csr
            res = shard.cpu().numpy().reshape(-1, 3)[:int(ro[-1])]
            n_ref = 2000 if O.lib_ref() is not None else n_check
            es = np.unique((np.arange(n_ref, dtype=np.int64) * 7919) % max(1, int(ro[-1])))
            rows_of = np.searchsorted(ro, es, side="right") - 1
            if O.lib_ref() is not None:
                want = O.ref_rows(seqs, rr[rows_of], np.arange(len(es) + 1, dtype=np.uint64), rq[es], params, threads=host_cores())
                ok = bool(np.array_equal(res[es], want))
            else:
                ok = all(tuple(int(x) for x in res[e]) == O.oracle_pair(seqs[int(rr[k])], seqs[int(rq[e])], params) for e, k in zip(es, rows_of))
            out["parity_on_last_slab"] = "bit-exact" if ok else "MISMATCH"
            out["parity_on_last_slab_pairs"] = len(es)
        elif world == 1 and not args.no_check:
            # pairs of the last timed slab, straight from the bench's own result buffer: >= 2,000 of them against the reference's
            # CParser (oracle/_ref) where it is built, else n_check against the C restatement
            import oracle as O
            res = shard.cpu().numpy().reshape(rows_max, n - 1, 3)
            n_ref = 0 if O.lib_ref() is None else max(2000, n_check)
            picks = []
            for k in range(max(n_ref, n_check)):
                i = (k * 7919) % len(mine)
                r = int(mine[i])
                qq = (r + 1 + (k * 104729) % (n - 1)) % n
                picks.append((i, r, qq))
            if n_ref:
                rr = np.array([p[1] for p in picks], np.uint32)
                want = O.ref_rows(seqs, rr, np.arange(len(picks) + 1, dtype=np.uint64), np.array([p[2] for p in picks], np.uint32), params,
                                  threads=host_cores())
                gotp = np.stack([res[i, qq if qq < r else qq - 1] for i, r, qq in picks])
                ok = bool(np.array_equal(gotp, want))
            else:
                ok = all(tuple(int(x) for x in res[i, qq if qq < r else qq - 1]) == O.oracle_pair(seqs[r], seqs[qq], params) for i, r, qq in picks)
            out["parity_on_last_slab"] = "bit-exact" if ok else "MISMATCH"
            out["parity_on_last_slab_pairs"] = len(picks)
        else:
            if world > 1:
                # N > 1: pairs from every rank's rows of the last gathered slab against the oracle
                import oracle as O
                g = gathered.cpu().numpy().reshape(world, rows_max, n - 1, 3)
                ok = True
                for k in range(n_check):
                    rk = k % world
                    theirs = SH.rank_rows(rows, rk, world)
                    if not len(theirs):
                        continue
                    i = (k * 7919) % len(theirs)
                    r = int(theirs[i])
                    qq = (r + 1 + (k * 104729) % (n - 1)) % n
                    ok &= tuple(int(x) for x in g[rk, i, qq if qq < r else qq - 1]) == O.oracle_pair(seqs[r], seqs[qq], params)
                out["parity_on_last_slab"] = "bit-exact" if ok else "MISMATCH"
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    bad = out is not None and "MISMATCH" in (out.get("parity_on_last_slab"), (out.get("cpu_baseline") or {}).get("parity_on_sample"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
