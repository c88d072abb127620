#!/usr/bin/env python3
"""GPU box: the filtered (all pairs related) shape of the workload - what a kmer-db prefilter leaves.
n genomes in families of `fam`; the rows hold the same-family pairs only.  Prints the GPU rate and the
reference's rate (oracle/_ref, all host threads) on the same rows.  Usage: tools/related_bench.py [n] [fam] [dmax]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("lz-ani_amd", "oracle", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import lzani_ctypes as L
import oracle as O
import synth_genomes as SG

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
fam = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dmax = float(sys.argv[3]) if len(sys.argv) > 3 else 0.15
_, seqs = SG.make_set(n, 1, fam=fam, dmax=dmax)
rows = [[q for q in range((r // fam) * fam, min(n, (r // fam + 1) * fam)) if q != r] for r in range(n)]
ref_ids = np.arange(n, dtype=np.uint32)
row_off = np.zeros(n + 1, dtype=np.uint64)
row_off[1:] = np.cumsum([len(r) for r in rows])
q = np.array([x for r in rows for x in r], dtype=np.uint32)
eng = L.Engine()
eng.set_genomes(seqs)
eng.run_rows(ref_ids, row_off, q)
t = time.perf_counter(); got = eng.run_rows(ref_ids, row_off, q); t_gpu = time.perf_counter() - t
tm = eng.timing()
eng.close()
threads = len(os.sched_getaffinity(0))
try:
    a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if a != "max": threads = max(1, min(threads, -(-int(a) // int(b))))
except Exception:
    pass
m = min(n, 8 * fam)                                    # CPU sample: the first families
e1 = int(row_off[m])
t = time.perf_counter(); want = O.ref_rows(seqs, ref_ids[:m], row_off[:m + 1], q[:e1], None, threads); t_cpu = time.perf_counter() - t
bad = int((got[:e1] != want).any(axis=1).sum())
print(f"{n} genomes, families of {fam}, divergence <= {dmax}: {len(q)} related pairs; GPU {t_gpu*1e3:.1f} ms wall "
      f"(pair kernel {tm['pairs_ms']:.1f} ms, index {tm['index_ms']:.1f} ms) = {len(q)/t_gpu:.0f} pairs/s; reference on {threads} threads: "
      f"{e1} pairs in {t_cpu:.2f} s = {e1/t_cpu:.0f} pairs/s; differing pairs in the sample: {bad}")
sys.exit(1 if bad else 0)
