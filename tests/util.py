"""Shared helpers of the test-suite: fixture loading, seeded edge cases, the TSV emit rule."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

import oracle as O
import synth_genomes as SG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

VARIANTS = {
    "default": {},
    "long": dict(mal=15, msl=9, reg=60),                  # BASELINE config 4 parameters
    "mrd20_mqd60": dict(mrd=20, mqd=60),
    "aw10_am3_ar5": dict(aw=10, am=3, ar=5),
    "short": dict(mrd=10, mqd=10, mal=9, msl=5, reg=20),
}


def load_example():
    recs = O.read_multifasta(os.path.join(GOLD, "example", "multifasta.fna"))
    return [r[0] for r in recs], [r[1] for r in recs]


def load_vir61():
    """--in-dir semantics: files sorted by name, every '>' record one item (multisample default)."""
    names, seqs = [], []
    for f in sorted(glob.glob(os.path.join(GOLD, "vir61", "*.fna"))):
        for nm, s in O.read_multifasta(f):
            names.append(nm)
            seqs.append(s)
    return names, seqs


def reorder(names, seqs):
    """CSeqReservoir::reorder_items (seq_reservoir.cpp:215-251): (len - 2*no_parts) as uint32, descending,
    then name ascending (bytewise); stable."""
    key = [((len(s) - 2) & 0xFFFFFFFF) for s in seqs]
    order = sorted(range(len(seqs)), key=lambda i: (-key[i], names[i].encode()))
    return [names[i] for i in order], [seqs[i] for i in order]


def edge_set():
    """Seeded corner cases: copies, reverse complement, N runs, SNPs, tiny and empty inputs, repeats."""
    st = SG.Stream(5)
    base = (st.u64(3000) % np.uint64(4)).astype(np.uint8)
    rc = (3 - base[::-1]).astype(np.uint8)
    snp = base.copy()
    snp[::50] = (snp[::50] + 1) % 4
    nins = np.concatenate([base[:1500], np.full(50, 5, np.uint8), base[1500:]])
    mosaic = np.concatenate([base[100:900], rc[1000:2000], base[2100:2500]])
    return [base, base.copy(), rc, nins, snp, base[:1500].copy(), np.full(200, 5, np.uint8),
            base[:12].copy(), base[:5].copy(), np.zeros(0, np.uint8), np.zeros(500, np.uint8),
            np.tile(np.array([0, 1], np.uint8), 400), mosaic,
            np.concatenate([np.full(3, 5, np.uint8), base[:700], np.full(1, 5, np.uint8), base[700:1400]])]


# ---- the reference's number formatting and TSV emit rule (lz_matcher.cpp:280-579,
# ---- numeric_conversions.h:228-300), restated for the tests -------------------------------
def real_to_str(v, prec):
    if v == 0:
        return "0"
    r = repr(float(v))
    mant, _, ex = r.partition("e")
    exp10 = int(ex) if ex else 0
    if "." in mant:
        ip, fp = mant.split(".")
    else:
        ip, fp = mant, ""
    digits = (ip + fp).lstrip("0")
    exp10 -= len(fp)
    lead_stripped = len(ip + fp) - len((ip + fp).lstrip("0"))
    del lead_stripped
    t = digits.rstrip("0")
    exp10 += len(digits) - len(t)
    sig = int(t)
    nd = len(t)
    if nd > prec:
        p10 = 10 ** (nd - prec)
        sig = (sig + p10 // 2) // p10
        exp10 += nd - prec
        nd = prec
        if sig >= 10 ** prec:
            sig //= 10
            exp10 += 1
    s = str(sig)
    if exp10 == 0:
        return s
    if exp10 > 0 or -exp10 >= nd + 4:
        e = exp10
        if nd == 1:
            out = s
        else:
            out = s[0] + "." + s[1:]
            e += nd - 1
        return out + ("e-%02d" % -e if e < 0 else "e+%02d" % e)
    if -exp10 < nd:
        k = nd + exp10
        return s[:k] + "." + s[k:]
    return "0." + "0" * (-exp10 - nd) + s


def emit_tsv(names, lens, res, columns, in_percent=False):
    """store_results for dense results res[r, q] (ids already in reordered order)."""
    mult = 100.0 if in_percent else 1.0
    lines = ["\t".join(columns)]
    n = len(names)
    for a in range(n):
        for b in range(a + 1, n):
            X = res[a, b]   # parse(query=b, ref=a)
            Y = res[b, a]   # parse(query=a, ref=b)
            ids = (a, b)
            ln = (lens[b], lens[a])
            mat = (int(X[0]), int(Y[0]))
            lit = (int(X[1]), int(Y[1]))
            reg = (int(X[2]), int(Y[2]))
            tani = (mat[0] + mat[1]) / (ln[0] + ln[1])
            gani = (mat[0] / ln[0], mat[1] / ln[1])
            ani = tuple(m / (m + l) if m + l else 0.0 for m, l in zip(mat, lit))
            cov = ((mat[0] + lit[0]) / ln[0], (mat[1] + lit[1]) / ln[1])
            for i in (0, 1):
                j = 1 - i
                f = {"ridx": str(ids[i]), "qidx": str(ids[j]), "reference": names[ids[i]], "query": names[ids[j]],
                     "qcov": real_to_str(mult * cov[i], 6), "rcov": real_to_str(mult * cov[j], 6),
                     "gani": real_to_str(mult * gani[i], 6), "ani": real_to_str(mult * ani[i], 6),
                     "tani": real_to_str(mult * tani, 6), "rlen": str(ln[j]), "qlen": str(ln[i]),
                     "num_alns": str(reg[i]), "nt_match": str(mat[i]), "nt_mismatch": str(lit[i])}
                if ln[0] and ln[1]:
                    f["len_ratio"] = real_to_str(min(ln[i], ln[j]) / max(ln[i], ln[j]), 4)
                else:
                    f["len_ratio"] = "0"
                lines.append("\t".join(f[c] for c in columns))
    return "\n".join(lines) + "\n"


STANDARD = "qidx,ridx,query,reference,tani,gani,ani,qcov,num_alns,len_ratio".split(",")


# ---- the lane-emulating host model of the kernels (tests/model) --------------------------
_model = None


def model_lib():
    global _model
    if _model is None:
        d = os.path.join(ROOT, "tests", "model")
        so = os.path.join(d, "liblzani_model.so")
        src = os.path.join(d, "lzani_model.cpp")
        hdrs = [os.path.join(ROOT, "lz-ani_amd", "csrc", h) for h in ("lzani_core.h", "lzani_layout.h")]
        if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in [src] + hdrs):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", so])
        _model = C.CDLL(so)
    return _model


def model_all2all(seqs, params=None):
    lib = model_lib()
    seqs, ptrs, lens = O._seq_table(seqs)
    n = len(seqs)
    out = np.zeros((n, n, 3), dtype=np.int32)
    rc = lib.model_all2all(n, ptrs, O._ptr(lens), O.params_array(params), O._ptr(out))
    if rc != 0:
        raise ValueError("model: unsupported parameters")
    return out


def model_split_all2all(seqs, params=None, seglen=1000):
    """Every pair by several segments (lzani_core.h: checkpoints, segments, stitch) through the host model: (results, stats)
    with stats = [pairs stitched, pairs the stitch voided, segments in all, segments skipped by the hand-overs]."""
    lib = model_lib()
    seqs, ptrs, lens = O._seq_table(seqs)
    n = len(seqs)
    out = np.zeros((n, n, 3), dtype=np.int32)
    stats = np.zeros(4, dtype=np.int64)
    rc = lib.model_split_all2all(n, ptrs, O._ptr(lens), O.params_array(params), int(seglen), O._ptr(out), O._ptr(stats))
    if rc != 0:
        raise ValueError("model: unsupported parameters")
    return out, stats


def model_pair_regions(ref, qry, params=None):
    """(result triple, regions sorted like calc_regions) through the ALN instantiation of the model."""
    lib = model_lib()
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    qry = np.ascontiguousarray(qry, dtype=np.uint8)
    res = np.zeros(3, dtype=np.int32)
    regs = np.zeros((1 << 14, 6), dtype=np.int32)
    n = C.c_uint32(0)
    rc = lib.model_pair_regions(O._ptr(ref), len(ref), O._ptr(qry), len(qry), O.params_array(params), O._ptr(res),
                                O._ptr(regs), 1 << 14, C.byref(n))
    if rc != 0:
        raise ValueError("model: unsupported parameters")
    g = regs[:n.value]
    order = sorted(range(len(g)), key=lambda k: (-(int(g[k][3]) - int(g[k][2])), int(g[k][2])))
    return tuple(int(x) for x in res), g[order]


DEFAULTS = dict(mal=11, msl=7, mrd=40, mqd=40, reg=35, aw=15, am=7, ar=3)


def fuzz_case_medium(st):
    """Like fuzz_case, at genome sizes where the tag words, the bucket table and the LDS index build are in
    use (8-70 kbp, mal <= 12): an ancestor, two mutated copies (one with an inversion or N runs), one stranger."""
    msl = st.randint(4, 9)
    mal = st.randint(max(msl, 9), 12)
    mrd = st.randint(8, 64)
    prm = dict(mal=mal, msl=msl, mrd=mrd, mqd=st.randint(4, min(mrd, 64)), reg=st.randint(10, 80), aw=st.randint(4, 40),
               am=st.randint(1, 12), ar=st.randint(1, 6))
    pick = st.one()
    if pick < 0.3:
        prm = dict(DEFAULTS)
    elif pick < 0.45:                          # the second parameter set the pair kernel folds into its code (null chain included)
        prm = dict(DEFAULTS, **VARIANTS["long"])
    L = st.randint(8000, 70000)
    base = (st.u64(L) % np.uint64(4)).astype(np.uint8)
    seqs = [base, SG.mutate(base, 0.01 + 0.12 * st.one(), st)]
    g = SG.mutate(base, 0.02 + 0.2 * st.one(), st).copy()
    if st.one() < 0.5:
        for _ in range(st.randint(1, 3)):
            a = st.randint(0, len(g) - 200)
            g[a:a + st.randint(1, 120)] = 5
    else:
        a = st.randint(0, len(g) - 3000)
        g[a:a + 2500] = (3 - g[a:a + 2500][::-1])
    seqs.append(np.ascontiguousarray(g))
    seqs.append((st.u64(st.randint(8000, 70000)) % np.uint64(4)).astype(np.uint8))
    return prm, seqs


def fuzz_params_chain(st):
    """A random parameter tuple INSIDE what the hand-written null chain is written for (chain_params_ok in
    lzani_kernels_pairs.h: mqd <= 63, 64 <= mqd + mrd <= 128, 2 <= aw <= 15, ar <= aw, msl <= 9) and inside the reference's
    own well-defined range (mqd <= mrd); never one of the two tuples compiled ahead of time."""
    while True:
        mrd = st.randint(32, 64)
        mqd = st.randint(64 - mrd, min(mrd, 63))
        msl = st.randint(4, 9)
        aw = st.randint(2, 15)
        prm = dict(mal=st.randint(max(msl, 9), 13), msl=msl, mrd=mrd, mqd=mqd, reg=st.randint(10, 80), aw=aw,
                   am=st.randint(0, aw), ar=st.randint(0, min(aw, 6)))
        if prm != DEFAULTS and prm != dict(DEFAULTS, **VARIANTS["long"]):
            return prm


def fuzz_seqs_medium(st, with_n=None):
    """The sequences of fuzz_case_medium (an ancestor, two mutated copies, a stranger; 8-70 kbp); with_n: False = no N
    anywhere (the N-free kernel instantiation), True = N runs in one copy, None = either."""
    L = st.randint(8000, 70000)
    base = (st.u64(L) % np.uint64(4)).astype(np.uint8)
    seqs = [base, SG.mutate(base, 0.01 + 0.12 * st.one(), st)]
    g = SG.mutate(base, 0.02 + 0.2 * st.one(), st).copy()
    n_runs = st.one() < 0.5 if with_n is None else with_n
    if n_runs:
        for _ in range(st.randint(1, 3)):
            a = st.randint(0, len(g) - 200)
            g[a:a + st.randint(1, 120)] = 5
    else:
        a = st.randint(0, len(g) - 3000)
        g[a:a + 2500] = (3 - g[a:a + 2500][::-1])
    seqs.append(np.ascontiguousarray(g))
    seqs.append((st.u64(st.randint(8000, 70000)) % np.uint64(4)).astype(np.uint8))
    return seqs


def fuzz_case_large(st):
    """Like fuzz_case_medium at 0.3-1.2 Mbp: directories of 2^20 buckets and more (the sort-based index build) and, from
    ~0.5 Mbp on, tag words of 8 MB (candidates by the join): an ancestor, a mutated copy, a stranger."""
    prm, _ = fuzz_case_medium(st)
    L = st.randint(300_000, 1_200_000)
    base = (st.u64(L) % np.uint64(4)).astype(np.uint8)
    g = SG.mutate(base, 0.01 + 0.1 * st.one(), st).copy()
    if st.one() < 0.5:
        a = st.randint(0, len(g) - 500)
        g[a:a + st.randint(1, 300)] = 5
    other = (st.u64(st.randint(300_000, 1_200_000)) % np.uint64(4)).astype(np.uint8)
    return prm, [base, np.ascontiguousarray(g), other]


def fuzz_case(st):
    """One random differential case: LZ parameters inside the engine's envelope with mqd <= mrd (beyond
    that the reference reads past the end of its reference text, parser.cpp:288/713, and its answer
    depends on stale heap bytes), and five short genomes: an ancestor, mutated copies, unrelated
    sequences, N runs, reverse complements, low-complexity repeats."""
    wide = st.one() < 0.2                      # long k-mers / wide seed windows: the engine's generic paths
    msl = st.randint(1, 20 if wide else 12)
    mal = st.randint(msl, min(28 if wide else 16, msl + 8))
    mrd = st.randint(0, 300 if wide else 64)
    prm = dict(mal=mal, msl=msl, mrd=mrd, mqd=st.randint(0, min(mrd, 64)), reg=st.randint(1, 80), aw=st.randint(1, 64),
               am=st.randint(0, 20), ar=st.randint(0, 12))
    base = (st.u64(st.randint(50, 1500)) % np.uint64(4)).astype(np.uint8)
    seqs = [base]
    for _ in range(4):
        if st.one() < 0.8:
            g = SG.mutate(base, 0.01 + 0.25 * st.one(), st)
        else:
            g = (st.u64(st.randint(20, 1500)) % np.uint64(4)).astype(np.uint8)
        if st.one() < 0.3:
            g = g.copy()
            a = st.randint(0, max(0, len(g) - 10))
            g[a:a + st.randint(1, 30)] = 5
        if st.one() < 0.2:
            g = (3 - g[::-1]).astype(np.uint8)
            g[g > 3] = 5
        if st.one() < 0.15:
            g = np.tile(g[:st.randint(1, 6)], 60)[:400]
        seqs.append(np.ascontiguousarray(g))
    return prm, seqs
