#!/usr/bin/env python3
"""Long differential fuzz of the HIP path against the oracle.  Usage: tools/fuzz_gpu.py [seed] [seconds] [medium|large]
`medium`: 8-70 kbp genomes (tag words, bucket table, LDS index build in use) instead of the tiny ones;
`large`: 0.3-1.2 Mbp (sort-based index build, join form of candidate detection);
`rtc [cases_per_tuple]`: random parameter TUPLES through the pair kernels compiled at run time for them (lzani_rtc.h) --
three tuples out of four inside the null chain's envelope -- cases_per_tuple (default 50) sequence sets of 8-70 kbp each, with
and without N, dense rows by candidate bitmaps and filtered rows by probes (or bitmaps, where the index has no tag words); every launch of the
anchor-queue kernels must be a run-time compiled one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("lz-ani_amd", "oracle", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import lzani_ctypes as L
import oracle as O
import synth_genomes as SG
import util as U

if len(sys.argv) > 3 and sys.argv[3] == "rtc":
    os.environ["LZANI_RTC_MIN_PAIRS"] = "0"
    st = SG.Stream(int(sys.argv[1]))
    budget = float(sys.argv[2])
    per = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    t0, tuples, cases, bad, launches, chain_tuples, build_ms = time.time(), 0, 0, 0, 0, 0, 0.0
    while time.time() - t0 < budget:
        prm = U.fuzz_params_chain(st) if st.one() < 0.75 else U.fuzz_case_medium(st)[0]
        if prm == U.DEFAULTS or prm == dict(U.DEFAULTS, **U.VARIANTS["long"]):
            continue
        tuples += 1
        for k in range(per):
            seqs = U.fuzz_seqs_medium(st, with_n=(k % 3 == 2))
            want = O.oracle_all2all(seqs, prm, threads=16)
            eng = L.Engine(prm)
            eng.set_genomes(seqs)
            os.environ["LZANI_PM_MIN_ROWS"] = "1"
            got = eng.all2all()                                   # dense rows: candidate bitmaps
            del os.environ["LZANI_PM_MIN_ROWS"]
            la = eng.layout()
            n = len(seqs)
            ref_ids = np.arange(n, dtype=np.uint32)
            qs = [[x for x in range(n) if x != r] for r in range(n)]
            row_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(n - 1)
            out = eng.run_rows(ref_ids, row_off, np.array(qs, np.uint32).reshape(-1)).reshape(n, n - 1, 3)   # query lists: probes
            lb = eng.layout()
            info = eng.rtc_info()
            eng.close()
            launches += la["rtc_launches"] + lb["rtc_launches"]
            if k == 0:
                chain_tuples += info["null_chain"]
            build_ms += info["build_ms"]
            ok = np.array_equal(got, want) and all(np.array_equal(out[r], want[r, qs[r]]) for r in range(n))
            if not ok or la["rtc_launches"] != la["bitmap_launches"] or la["bitmap_launches"] != 1 or lb["rtc_launches"] != max(lb["tag_words"], lb["bitmap_launches"]):
                bad += 1
                print("RTC MISMATCH" if not ok else "NOT THE RUN-TIME KERNEL", prm, [len(x) for x in seqs], la, lb, info, flush=True)
            cases += 1
        print("... tuple", tuples, prm, "cases", cases, "bad", bad, "%.0f s" % (time.time() - t0), flush=True)
    print("rtc fuzz: tuples", tuples, "(inside the chain's envelope:", chain_tuples, ") cases", cases, "mismatches", bad,
          "run-time compiled launches", launches, "build+load ms in all %.0f" % build_ms)
    sys.exit(1 if bad else 0)

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
