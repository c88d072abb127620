#!/usr/bin/env python3
"""Long differential fuzz of the HIP path against the oracle.  Usage: tools/fuzz_gpu.py [seed] [seconds] [medium|large]
`medium`: 8-70 kbp genomes (tag words, bucket table, LDS index build in use) instead of the tiny ones;
`large`: 0.3-1.2 Mbp (sort-based index build, join form of candidate detection)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("lz-ani_amd", "oracle", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import lzani_ctypes as L
import oracle as O
import synth_genomes as SG
import util as U

st = SG.Stream(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
medium = len(sys.argv) > 3 and sys.argv[3] == "medium"
large = len(sys.argv) > 3 and sys.argv[3] == "large"
t0, it, bad, blk, pmx = time.time(), 0, 0, 0, 0
while time.time() - t0 < budget:
    it += 1
    prm, seqs = U.fuzz_case_large(st) if large else U.fuzz_case_medium(st) if medium else U.fuzz_case(st)
    eng = L.Engine(prm)
    eng.set_genomes(seqs)
    got = eng.all2all()
    want = O.oracle_all2all(seqs, prm, threads=16 if (medium or large) else 4)
    if not np.array_equal(got, want):
        bad += 1
        print("MISMATCH", prm, [len(s) for s in seqs], np.argwhere((got != want).any(axis=2))[:3].tolist(), flush=True)
    if medium or large:                               # dense rows with their candidates from the presence matrix (by default from 32 rows on)
        os.environ["LZANI_PM_MIN_ROWS"] = "1"
        got2 = eng.all2all()
        del os.environ["LZANI_PM_MIN_ROWS"]
        pmx += eng.layout()["bitmap_launches"]
        if not np.array_equal(got2, want):
            bad += 1
            print("PM MISMATCH", prm, [len(s) for s in seqs], np.argwhere((got2 != want).any(axis=2))[:3].tolist(), flush=True)
    if medium:                                        # rows of >= 128 pairs (every query 45 times): the block kernel with the LDS filter
        n = len(seqs)
        ref_ids = np.arange(n, dtype=np.uint32)
        qs = [[x for x in range(n) if x != r] * 45 for r in range(n)]
        row_off = np.zeros(n + 1, np.uint64)
        row_off[1:] = np.cumsum([len(x) for x in qs])
        qq = np.array([x for row in qs for x in row], np.uint32)
        os.environ["LZANI_BLOCK_KERNEL"] = "1"           # (filtered rows take the wave kernel by default)
        out = eng.run_rows(ref_ids, row_off, qq).reshape(-1, 3)
        del os.environ["LZANI_BLOCK_KERNEL"]
        blk += eng.layout()["block_launches"]
        wantr = np.concatenate([want[r, qs[r]] for r in range(n)])
        if not np.array_equal(out, wantr):
            bad += 1
            print("ROWS MISMATCH", prm, [len(s) for s in seqs], np.argwhere((out != wantr).any(axis=1))[:3].tolist(), flush=True)
    if it % 4 == 0 and not large:                     # the alignment instantiation: regions of one row (the oracle wrapper holds 4,096 regions per pair: not at Mbp sizes)
        n = len(seqs)
        r = it % n
        ref_ids, row_off = L.dense_rows(n, [r])
        out, regs = eng.run_rows_regions(ref_ids, row_off, None, capacity=256)
        e = 0
        for q in range(n):
            if q == r:
                continue
            mine = regs[regs["pair"] == e]
            cols = ("ref_start", "ref_end", "seq_start", "seq_end", "num_matches", "num_mismatches")
            g = np.stack([mine[k] for k in cols], axis=1) if len(mine) else np.zeros((0, 6), np.int32)
            ores, oregs = O.oracle_pair(seqs[r], seqs[q], prm, want_regions=True)
            if tuple(out[e]) != ores or not np.array_equal(g, oregs):
                bad += 1
                print("REGIONS MISMATCH", prm, [len(s) for s in seqs], r, q, flush=True)
            e += 1
    eng.close()
    if it % 500 == 0:
        print("...", it, "cases", flush=True)
print("cases", it, "mismatches", bad, "block-kernel launches", blk, "bitmap-fed launches", pmx)
sys.exit(1 if bad else 0)
