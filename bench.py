#!/usr/bin/env python3
"""bench.py -- all2all ANI hot path on MI355X: directed genome pairs per second.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1: under torch.distributed.run)
prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1]): synthetic viral genomes, ~40 kbp,
families of 10 with 1-15 % divergence, default LZ parameters, dense all2all.  At N = 1 the set has
1,000 genomes = 999,000 directed pairs per step.  At N > 1 the rows of the all2all (one row = one
reference against every other genome) are sharded cyclically over the ranks, every rank keeps the
whole packed genome set resident, and the per-pair results are gathered with one RCCL all_gather;
the set grows as 1000*sqrt(N) genomes so that each GPU keeps ~10^6 pairs ("weak" scaling).

A step = one pass of the hot path over the rank's rows: per-reference index build + pair kernel
(+ the gather when N > 1), genomes already resident in HBM, results left in HBM.
`roofline` is for the pair kernel (k_pairs): algorithmic bytes B_pair summed over the
pairs of a launch, divided by the launch duration measured with HIP events on the engine's stream.
`cpu_baseline` times the reference's own CParser (oracle/_ref, kind "reference"; falls back to the
C restatement, kind "port") on a bounded sample of the same workload, rank 0, N = 1 only.
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


def algorithmic_bytes(lens, ref_ids, params):
    """Sum over the dense rows `ref_ids` of B_pair = ceil(Lq/4) + ceil((2Lr+3mrd)/4) + 4(Lq+mrd-mal+1) + 12."""
    lens = lens.astype(np.int64)
    n = len(lens)
    mrd, mal = params["mrd"], params["mal"]
    q_bytes = (lens + 3) // 4 + 4 * np.maximum(lens + mrd - mal + 1, 0) + 12
    total_q = int(q_bytes.sum())
    tot = 0
    for r in ref_ids:
        r = int(r)
        tot += (n - 1) * int((2 * lens[r] + 3 * mrd + 3) // 4) + (total_q - int(q_bytes[r]))
    return tot


def pmc_traffic(n, seed, world):
    """HBM bytes per k_pairs launch from the committed rocprofv3 PMC passes (profiles/latest_pmc.json),
    quoted only when they were collected on this very workload; PMC cannot be read from inside the run."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_pmc.json")) as f:
            rec = json.load(f)
        w = rec["workload"]
        if world == 1 and w["genomes"] == n and w["seed"] == seed:
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


def cpu_baseline(seqs, params, sample):
    """Reference CParser (or the C port) on the dense all2all of the first `sample` genomes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O
    cores = host_cores()
    sub = seqs[:sample]
    npairs = len(sub) * (len(sub) - 1)
    if O.lib_ref() is not None:
        kind = "reference"
        t = time.perf_counter()
        res = O.ref_all2all(sub, params, threads=cores)
        dt = time.perf_counter() - t
    else:
        kind = "port"
        t = time.perf_counter()
        res = O.oracle_all2all(sub, params, threads=cores)
        dt = time.perf_counter() - t
    return dict(value=npairs / dt, unit="genome-pairs/s", cores=cores, kind=kind,
                sample=f"dense all2all of the first {len(sub)} genomes of the workload ({npairs} pairs, {dt:.2f} s wall, "
                       f"{cores} threads self-scheduling over reference rows as in do_matching)"), res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--genomes", type=int, default=0, help="0 = 1000*sqrt(gpus)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=192, help="genomes in the CPU baseline sample (0 = skip)")
    ap.add_argument("--lmin", type=int, default=36000, help="ancestor length range of the synthetic set")
    ap.add_argument("--lmax", type=int, default=44000)
    ap.add_argument("--params", default="", help="LZ parameter overrides, e.g. mal=15,msl=9,reg=60 (BASELINE configs[3])")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for 1-GPU rehearsals)")
    ap.add_argument("--device", type=int, default=-1, help="force the HIP device ordinal (rehearsals: all ranks on GPU 0)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = local_rank if args.device < 0 else args.device
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=args.backend)

    n = args.genomes or int(round(1000 * math.sqrt(max(world, 1))))
    over = {k: int(v) for k, v in (kv.split("=") for kv in args.params.split(",") if kv)}
    names, seqs = SG.make_set(n, args.seed, lmin=args.lmin, lmax=args.lmax)
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    # reference order: length-descending, then name (CSeqReservoir::reorder_items, seq_reservoir.cpp:215-251)
    order = sorted(range(n), key=lambda i: (-int(lens[i]), names[i]))
    seqs = [seqs[i] for i in order]
    lens = lens[order]

    eng = L.Engine(over or None, device=dev)
    params = eng.params
    eng.set_genomes(seqs)                      # untimed: genomes resident in HBM before the timed region

    my_rows = SH.row_shard(n, rank, world)                        # cyclic row shard
    ref_ids, row_off = L.dense_rows(n, my_rows)
    my_pairs = int(row_off[-1])
    shard = torch.zeros(SH.shard_len(n, world), dtype=torch.int32, device="cuda")
    gathered = torch.zeros(world * shard.numel(), dtype=torch.int32, device="cuda") if world > 1 else None

    def step():
        eng.run_rows_device(ref_ids, row_off, None, shard.data_ptr())
        if world > 1:
            if args.backend == "nccl":
                dist.all_gather_into_tensor(gathered, shard)
            else:
                dist.all_gather(list(gathered.view(world, -1).unbind(0)), shard)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    kernel_ms, index_ms = 0.0, 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        tm = eng.timing()
        kernel_ms += tm["pairs_ms"]
        index_ms += tm["index_ms"]
        launches = tm["pair_launches"]
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    total_pairs = n * (n - 1)
    out = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = total_pairs * args.steps / dt
        abytes = algorithmic_bytes(lens, my_rows, params)        # rank 0's launches
        avg_launch_ms = kernel_ms / max(1, args.steps * launches)
        achieved = abytes / launches / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        out = {
            "metric": "genome-pairs/sec + achieved HBM GB/s, 10k×40kbp all2all at 1/2/4/8 GPUs",
            "value": value, "unit": "genome-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 (2-bit packed symbols, 64-bit lane masks; f64 only in the anchor/seed arbitration)",
            "data": "synthetic",
            "config": {"workload": f"{n} synthetic genomes of {args.lmin}-{args.lmax} bp (families of 10, 1-15% divergence, seed {args.seed}), "
                                   f"dense all2all, {'default LZ params' if not over else 'LZ params ' + args.params}, {total_pairs} directed pairs/step",
                       "genomes": n, "pairs_per_step": total_pairs, "pairs_per_gpu": my_pairs,
                       "sharding": "reference rows cyclic over ranks, genomes replicated, one RCCL all_gather of int32[3] per pair"
                                   if world > 1 else "single GPU, all rows",
                       "params": params},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(n, args.seed, world),
                         "kernel": "k_pairs", "avg_launch_ms": avg_launch_ms, "launches_per_step": launches,
                         "algorithmic_bytes_per_launch": abytes / launches,
                         "index_build_ms_per_step": index_ms / args.steps},
        }
        if world == 1 and args.cpu_sample > 1:
            cb, cpu_res = cpu_baseline(seqs, params, min(args.cpu_sample, n))
            out["cpu_baseline"] = cb
            # free parity evidence: the sampled pairs, GPU vs CPU
            m = cpu_res.shape[0]
            got = shard.cpu().numpy()[: my_pairs * 3].reshape(n, n - 1, 3)
            ok = True
            for r in range(m):
                qs = [q for q in range(m) if q != r]
                cols = [q if q < r else q - 1 for q in qs]
                ok &= bool(np.array_equal(got[r, cols], cpu_res[r, qs]))
            out["cpu_baseline"]["parity_on_sample"] = "bit-exact" if ok else "MISMATCH"
        else:
            out["cpu_baseline"] = None
            if world > 1:
                # N > 1: reassemble the gathered shards and spot-check pairs from every rank's rows
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import oracle as O
                res = SH.assemble(gathered.cpu().numpy(), n, world)
                ok = True
                for k in range(48):
                    r = (k * 7919 + k % world) % n
                    q = (r + 1 + (k * 104729) % (n - 1)) % n
                    ok &= tuple(int(x) for x in res[r, q]) == O.oracle_pair(seqs[r], seqs[q], params)
                out["parity_on_sample"] = "bit-exact" if ok else "MISMATCH"
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    bad = out is not None and "MISMATCH" in (out.get("parity_on_sample"), (out.get("cpu_baseline") or {}).get("parity_on_sample"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
