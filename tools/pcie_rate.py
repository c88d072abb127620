#!/usr/bin/env python3
"""GPU box: the PCIe-inclusive rate of the hot path -- lzani_run_rows with HOST result buffers (results cross PCIe inside the
timed call) on one slab of the bench workload.  Usage: tools/pcie_rate.py [n_genomes=10000] [seed=2] [slab_rows=500]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("lz-ani_amd", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import lzani_ctypes as L
import synth_genomes as SG

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2
slab = int(sys.argv[3]) if len(sys.argv) > 3 else 500
names, seqs = SG.make_set_cached(n, seed)
order = np.argsort([-len(s) for s in seqs], kind="stable")
seqs = [seqs[k] for k in order]                           # (the bench's and the reference's length-descending order)
eng = L.Engine()
t = time.perf_counter(); eng.set_genomes(seqs); t_set = time.perf_counter() - t
ref_ids = np.arange(slab, dtype=np.uint32)
row_off = (np.arange(slab + 1, dtype=np.uint64)) * np.uint64(n - 1)
eng.run_rows(ref_ids, row_off, None)
best = 1e9
for _ in range(3):
    t = time.perf_counter(); out = eng.run_rows(ref_ids, row_off, None); best = min(best, time.perf_counter() - t)
tm = eng.timing()
pairs = slab * (n - 1)
print(f"{n} genomes, slab of {slab} rows = {pairs} pairs: set_genomes {t_set*1e3:.0f} ms once; run_rows with host results {best*1e3:.1f} ms wall "
      f"(pair kernel {tm['pairs_ms']:.1f} ms + candidate stage {tm['cand_ms']:.1f} ms + index {tm['index_ms']:.1f} ms + {out.nbytes/1e6:.0f} MB of results over PCIe) "
      f"= {pairs/best/1e6:.2f} M pairs/s PCIe-inclusive")
eng.close()
