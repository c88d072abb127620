#!/usr/bin/env python3
"""Full-size parity run: every directed pair of a synthetic set, HIP path vs the reference's own CParser
(oracle/_ref) or, without it, the C restatement.  Usage: tools/full_parity.py [n_genomes] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("lz-ani_amd", "oracle", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import lzani_ctypes as L
import oracle as O
import synth_genomes as SG

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
_, seqs = SG.make_set(n, seed)
eng = L.Engine()
eng.set_genomes(seqs)
t = time.perf_counter(); got = eng.all2all(); t_gpu = time.perf_counter() - t
eng.close()
threads = len(os.sched_getaffinity(0))
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
    if q != "max": threads = max(1, min(threads, -(-int(q) // int(p))))
except Exception:
    pass
kind = "reference CParser (oracle/_ref)" if O.lib_ref() is not None else "C restatement"
t = time.perf_counter()
want = O.ref_all2all(seqs, None, threads=threads) if O.lib_ref() is not None else O.oracle_all2all(seqs, None, threads=threads)
t_cpu = time.perf_counter() - t
bad = int((got != want).any(axis=2).sum())
print(f"{n} genomes, {n*(n-1)} directed pairs: GPU {t_gpu:.2f} s, {kind} on {threads} threads {t_cpu:.1f} s "
      f"({n*(n-1)/t_cpu:.0f} pairs/s), differing pairs: {bad}", flush=True)
sys.exit(1 if bad else 0)
