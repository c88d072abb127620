"""GPU: parity of the HIP path (through the C-ABI) with the oracle and the committed golden vectors.
Bit-exact for every integer; the derived doubles are then byte-equal in the TSV."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import lzani_ctypes as L
import oracle as O
import synth_genomes as SG
import util as U

pytestmark = pytest.mark.gpu



def _bucketwise_sorted(dirz, ent):
    """Entries with every bucket sorted: the device leaves buckets of more than IDX_SORT_MAX entries in fill order."""
    out = ent.copy()
    big = np.nonzero(np.diff(dirz.astype(np.int64)) > 32)[0]
    for b in big:
        out[dirz[b]:dirz[b + 1]] = np.sort(out[dirz[b]:dirz[b + 1]])
    return out


def gpu_all2all(seqs, params=None):
    eng = L.Engine(params)
    try:
        eng.set_genomes(seqs)
        return eng.all2all()
    finally:
        eng.close()


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(U.GOLD, "ref_vectors.json")) as f:
        return json.load(f)


def _inputs(setname):
    return {"example": lambda: U.load_example()[1], "edge": U.edge_set, "vir61": lambda: U.load_vir61()[1],
            "synth24": lambda: SG.make_set(24, 11, lmin=6000, lmax=9000, fam=6)[1]}[setname]()


def test_reference_vectors_through_c_abi(vectors):
    """Every committed vector of the reference's CParser: example, edge set, synth24 under 12 parameter
    sets, vir61 (3,660 pairs) under defaults."""
    for key, item in vectors["sets"].items():
        got = gpu_all2all(_inputs(key.split("/")[0]), item["params"])
        want = np.array(item["res"], dtype=np.int32)
        bad = np.argwhere((got != want).any(axis=2))
        assert len(bad) == 0, f"{key}: {len(bad)} pairs differ, first {bad[:3].tolist()}: {got[tuple(bad[0])]} vs {want[tuple(bad[0])]}"


def test_vir61_golden_tsv_from_gpu_results():
    """BASELINE config 1 end to end on the GPU numbers: byte-identical to test/vir61.ani.tsv."""
    names, seqs = U.reorder(*U.load_vir61())
    res = gpu_all2all(seqs)
    txt = U.emit_tsv(names, [len(s) for s in seqs], res, U.STANDARD)
    assert txt == open(os.path.join(U.GOLD, "vir61.ani.tsv")).read()


def test_device_built_text_and_index_match_model():
    """k_pack and the k_idx_* kernels produce exactly the layout the host model builds."""
    _, seqs = U.load_example()
    seqs = seqs[:3] + [U.edge_set()[3], U.edge_set()[11]]
    lib = U.model_lib()
    for prm in (None, dict(mal=15, msl=9), dict(mal=5, msl=4)):
        eng = L.Engine(prm)
        eng.set_genomes(seqs)
        maxlen = max(len(s) for s in seqs)
        for gid, s in enumerate(seqs):
            d = eng.debug_index(gid)
            T = 2 * len(s) + 3 * eng.params["mrd"]
            wn = (T + 63) // 64 + 2
            t2 = np.zeros(2 * wn, np.uint64); nm = np.zeros(wn, np.uint64)
            dirz = np.zeros(len(d["dirz"]), np.uint32); ent = np.zeros(T + 1, np.uint32)
            n_ent = C.c_uint32(0)
            s = np.ascontiguousarray(s)
            assert lib.model_index(O._ptr(s), len(s), maxlen, O.params_array(prm), O._ptr(t2), O._ptr(nm),
                                   O._ptr(dirz), O._ptr(ent), C.byref(n_ent)) == 0
            assert np.array_equal(d["t2"], t2) and np.array_equal(d["nm"], nm)
            assert np.array_equal(d["dirz"], dirz)
            assert n_ent.value == len(d["ent"]) and np.array_equal(_bucketwise_sorted(d["dirz"], d["ent"]), ent[:n_ent.value])
        eng.close()


def test_sparse_rows_ragged_and_empty():
    """The filtered form of do_matching (lz_matcher.cpp:234-250): CSR rows with arbitrary query lists,
    including empty rows, repeated references and unsorted queries."""
    _, seqs = SG.make_set(20, 4, lmin=3000, lmax=6000, fam=5)
    want = O.oracle_all2all(seqs, None, threads=8)
    st = SG.Stream(99)
    ref_ids, row_off, q = [], [0], []
    for k in range(17):
        r = st.randint(0, 19)
        cnt = 0 if k % 5 == 2 else st.randint(1, 12)
        ref_ids.append(r)
        for _ in range(cnt):
            x = st.randint(0, 19)
            q.append(x if x != r else (x + 1) % 20)
        row_off.append(len(q))
    eng = L.Engine()
    eng.set_genomes(seqs)
    got = eng.run_rows(ref_ids, row_off, q)
    eng.close()
    k = 0
    for row, r in enumerate(ref_ids):
        for e in range(row_off[row], row_off[row + 1]):
            assert got[e].tolist() == want[r, q[e]].tolist(), (row, r, q[e])
            k += 1
    assert k == len(q)


def test_row_shards_equal_full_run_and_are_idempotent():
    _, seqs = SG.make_set(30, 8, lmin=3000, lmax=5000, fam=6)
    eng = L.Engine()
    eng.set_genomes(seqs)
    full = eng.all2all()
    again = eng.all2all()
    assert np.array_equal(full, again)
    n = len(seqs)
    for world in (2, 3):
        merged = np.zeros_like(full)
        for rank in range(world):
            rows = np.arange(rank, n, world, dtype=np.uint32)
            ref_ids, row_off = L.dense_rows(n, rows)
            flat = eng.run_rows(ref_ids, row_off, None).reshape(len(rows), n - 1, 3)
            for i, r in enumerate(rows):
                merged[r, np.arange(n) != r] = flat[i]
        assert np.array_equal(merged, full)
    eng.close()


def test_error_conventions():
    eng = L.Engine()
    with pytest.raises(L.LzaniError, match="LZANI_ERR_STATE"):
        eng.run_rows([0], [0, 0], None)
    seqs = U.edge_set()[:4]
    eng.set_genomes(seqs)
    with pytest.raises(L.LzaniError, match="LZANI_ERR_ARG"):
        eng.run_rows([9], [0, 3], None)                       # reference id out of range
    with pytest.raises(L.LzaniError, match="LZANI_ERR_ARG"):
        eng.run_rows([0], [0, 2], None)                       # dense row must have n-1 queries
    with pytest.raises(L.LzaniError, match="LZANI_ERR_ARG"):
        eng.run_rows([0], [0, 1], [7])                        # query id out of range
    assert eng.run_rows([], [0], None).shape == (0, 3)
    eng.close()
    with pytest.raises(L.LzaniError, match="LZANI_ERR_PARAMS"):
        L.Engine(dict(mqd=100))


def test_full_size_properties_1000_genomes():
    """BASELINE configs[1] at full size (1,000 x ~40 kbp, 999,000 directed pairs): size-independent
    properties on every pair plus exact agreement with the oracle on a seeded sample of pairs."""
    names, seqs = SG.make_set(1000, 1)
    lens = np.array([len(s) for s in seqs])
    eng = L.Engine()
    eng.set_genomes(seqs)
    res = eng.all2all()
    eng.close()
    mat, lit, comp = res[..., 0], res[..., 1], res[..., 2]
    assert (res >= 0).all()
    assert ((comp == 0) == ((mat == 0) & (lit == 0)))[~np.eye(1000, dtype=bool)].all()
    # a region needs >= reg symbols and cannot cover more than the query text
    assert (mat + lit >= 35 * comp).all()
    assert (mat + lit <= lens[None, :] + 40).all()
    # family structure: a genome shares most of its length with its own ancestor (member 0)
    fam0 = (np.arange(1000) // 10) * 10
    idx = np.arange(1000)
    rel = idx != fam0
    assert (mat[fam0[rel], idx[rel]] > 0.5 * lens[idx[rel]]).all()
    # unrelated random genomes share only chance matches (a few per cent at mal 11)
    other = (idx + 500) % 1000
    assert (mat[other, idx] < 0.05 * lens).all()
    # checksum of checksums is reproducible and matches the oracle on a seeded sample
    st = SG.Stream(123)
    pairs = [(st.randint(0, 999), st.randint(0, 999)) for _ in range(400)]
    pairs += [(10 * (k % 100), 10 * (k % 100) + 1 + k % 9) for k in range(200)]     # related pairs
    for r, q in pairs:
        if r != q:
            assert tuple(res[r, q]) == O.oracle_pair(seqs[r], seqs[q]), (r, q)


def test_long_genomes_config4_shape(monkeypatch):
    """BASELINE configs[3] in miniature: Mbp-scale genomes with --mal 15 --msl 9 --reg 60 (24-bit text
    positions, 2^21 bucket directory, several hundred extension chunks per match).  At this size the tags are 9
    bits wide -- no tag byte holds them, so there are no tag words and the probe form falls back to the rounds of the
    first kernel; dense rows (from 8 rows on) take their candidates from the presence matrix instead: the pair's
    bitmap, the candidate's bucket read whole.  Six genomes by rounds and, forced, by bitmaps; then 24 genomes of
    300 kbp (10-bit tags) both ways, timed."""
    import time
    _, seqs = SG.make_set(6, 3, lmin=900_000, lmax=1_100_000, fam=3, dmin=0.005, dmax=0.08)
    prm = dict(mal=15, msl=9, reg=60)
    want = O.oracle_all2all(seqs, prm, threads=16)
    eng = L.Engine(prm)
    eng.set_genomes(seqs)
    lay = eng.layout()
    assert lay["tag_words"] == 0 and lay["bucket_table"] == 1 and lay["join_lists"] == 0
    got = eng.all2all()
    assert eng.layout()["bitmap_launches"] == 0
    bad = np.argwhere((got != want).any(axis=2))
    assert len(bad) == 0, (bad[:4].tolist(), got[tuple(bad[0])], want[tuple(bad[0])])
    assert got[0, 1, 0] > 800_000          # related genomes really align over most of their length
    monkeypatch.setenv("LZANI_PM_MIN_ROWS", "1")
    got = eng.all2all()
    assert eng.layout()["bitmap_launches"] == 1
    monkeypatch.delenv("LZANI_PM_MIN_ROWS")
    eng.close()
    assert np.array_equal(got, want)
    _, seqs = SG.make_set(24, 5, lmin=280_000, lmax=320_000, fam=4, dmin=0.005, dmax=0.08)
    want = O.oracle_all2all(seqs, prm, threads=16)
    eng = L.Engine(prm)
    eng.set_genomes(seqs)
    assert eng.layout()["tag_words"] == 0
    got = eng.all2all()                       # (also makes the k-mer words: not in either timing below)
    assert eng.layout()["bitmap_launches"] == 1 and np.array_equal(got, want)
    t = time.perf_counter(); eng.all2all(); t_pm = time.perf_counter() - t
    monkeypatch.setenv("LZANI_PM", "0")
    t = time.perf_counter(); got0 = eng.all2all(); t_rounds = time.perf_counter() - t
    assert eng.layout()["bitmap_launches"] == 0
    eng.close()
    assert np.array_equal(got0, want)
    print(f"24 x 300 kbp, mal 15: {t_pm * 1e3:.0f} ms with candidate bitmaps, {t_rounds * 1e3:.0f} ms by rounds")


def test_filtered_heavy_tailed_rows_config5_shape():
    """BASELINE configs[4] in miniature: a symmetric sparse pair list with heavy-tailed row sizes
    (singletons, small and one large family plus random cross-family pairs), as a kmer-db prefilter gives."""
    st = SG.Stream(44)
    fams = [1] * 40 + [2] * 10 + [5] * 4 + [40]
    seqs, fam_of = [], []
    for f, size in enumerate(fams):
        anc = (st.u64(st.randint(2500, 4000)) % np.uint64(4)).astype(np.uint8)
        for m in range(size):
            seqs.append(anc if m == 0 else SG.mutate(anc, 0.02 + 0.1 * st.one(), st))
            fam_of.append(f)
    n = len(seqs)
    pairs = {(a, b) for a in range(n) for b in range(a + 1, n) if fam_of[a] == fam_of[b]}
    while len(pairs) < 900 + 120:
        a, b = st.randint(0, n - 1), st.randint(0, n - 1)
        if a != b:
            pairs.add((min(a, b), max(a, b)))
    rows = [[] for _ in range(n)]
    for a, b in sorted(pairs):
        rows[a].append(b)
        rows[b].append(a)
    ref_ids = np.arange(n, dtype=np.uint32)
    row_off = np.zeros(n + 1, dtype=np.uint64)
    row_off[1:] = np.cumsum([len(r) for r in rows])
    q = np.array([x for r in rows for x in r], dtype=np.uint32)
    eng = L.Engine()
    eng.set_genomes(seqs)
    got = eng.run_rows(ref_ids, row_off, q)
    eng.close()
    assert max(len(r) for r in rows) >= 39 and min(len(r) for r in rows) <= 2
    e = 0
    for r in range(n):
        for x in rows[r]:
            assert tuple(got[e]) == O.oracle_pair(seqs[r], seqs[x]), (r, x)
            e += 1


def test_regions_through_c_abi():
    """lzani_run_rows_regions (the --out-alignment path): every region of every example pair equals
    the oracle's calc_regions; the results_t triples are unchanged by the alignment instantiation."""
    _, ex = U.load_example()
    eng = L.Engine()
    eng.set_genomes(ex)
    n = len(ex)
    ref_ids, row_off = L.dense_rows(n)
    out, regs = eng.run_rows_regions(ref_ids, row_off, None, capacity=64)      # forces the grow-and-retry path
    plain = eng.run_rows(ref_ids, row_off, None)
    eng.close()
    assert np.array_equal(out, plain)
    e = 0
    total = 0
    for r in range(n):
        for q in range(n):
            if q == r:
                continue
            mine = regs[regs["pair"] == e]
            _, want = O.oracle_pair(ex[r], ex[q], None, want_regions=True)
            cols = ("ref_start", "ref_end", "seq_start", "seq_end", "num_matches", "num_mismatches")
            got = np.stack([mine[k] for k in cols], axis=1) if len(mine) else np.zeros((0, 6), np.int32)
            assert np.array_equal(got, want), (r, q)
            total += len(want)
            e += 1
    assert total == len(regs) and total > 100


def test_mixed_n_content_and_parameter_instantiations():
    """All kernel instantiations against the oracle on one set: genomes with and without N runs (NFREE
    on/off), default and non-default parameters (DEFP on/off), k-mer words on/off (mal 16 > 15)."""
    st = SG.Stream(77)
    _, seqs = SG.make_set(36, 9, lmin=5000, lmax=9000, fam=6)
    clean = [s.copy() for s in seqs]
    for k in range(0, 36, 3):                                  # a third of the genomes get N runs / IUPAC codes
        s = seqs[k].copy()
        for _ in range(st.randint(1, 4)):
            a = st.randint(0, len(s) - 60)
            s[a:a + st.randint(1, 50)] = 5
        seqs[k] = s
    for name, data in (("with N", seqs), ("N-free", clean)):
        for prm in (None, dict(reg=30), dict(mal=13, msl=8, mrd=30, mqd=50, aw=20, am=9, ar=2), dict(mal=16, msl=7)):
            got = gpu_all2all(data, prm)
            want = O.oracle_all2all(data, prm, threads=16)
            bad = np.argwhere((got != want).any(axis=2))
            assert len(bad) == 0, (name, prm, bad[:3].tolist())


@pytest.mark.parametrize("env", [{}, {"LZANI_NO_TAGWORDS": "1"}, {"LZANI_NO_BUCKETS": "1"}], ids=["tagwords", "buckets", "directory"])
def test_index_forms(monkeypatch, env):
    """The three forms of the anchor index the pair kernel can read (tag words + bucket table, bucket table
    alone, directory + entries) give the same results: genomes with and without N, default and other
    parameters, and the alignment instantiation."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _, seqs = SG.make_set(20, 11, lmin=17000, lmax=20000, fam=5)
    seqs[3] = np.concatenate([seqs[3][:4000], np.full(30, 5, np.uint8), seqs[3][4000:]])
    for data in (seqs, seqs[4:12]):                            # with the N genome / N-free
        for prm in (None, dict(reg=30, aw=20)):
            got = gpu_all2all(data, prm)
            want = O.oracle_all2all(data, prm, threads=16)
            assert np.array_equal(got, want), (env, prm)
    eng = L.Engine()
    eng.set_genomes(seqs[:8])
    ref_ids, row_off = L.dense_rows(8)
    out, regs = eng.run_rows_regions(ref_ids, row_off, None)
    eng.close()
    assert np.array_equal(out.reshape(-1, 3), O.oracle_all2all(seqs[:8], None, threads=16)[~np.eye(8, dtype=bool)])
    cols = ("ref_start", "ref_end", "seq_start", "seq_end", "num_matches", "num_mismatches")
    e = 0
    for r in range(8):
        for q in range(8):
            if q == r:
                continue
            mine = regs[regs["pair"] == e]
            _, want = O.oracle_pair(seqs[r], seqs[q], None, want_regions=True)
            got = np.stack([mine[k] for k in cols], axis=1) if len(mine) else np.zeros((0, 6), np.int32)
            assert np.array_equal(got, want), (env, r, q)
            e += 1


@pytest.mark.parametrize("env", [{}, {"LZANI_NO_LDS_INDEX": "1"}, {"LZANI_SORT_INDEX_MIN_DIRBITS": "0"}], ids=["lds-build", "global-atomics", "sort-build"])
def test_index_build_paths(monkeypatch, env):
    """k_idx_build (one block per reference through LDS) and its fallback: references the LDS build cannot take
    (a poly-A genome: one bucket holds every k-mer; a long tandem repeat; a genome with a 30 kbp low-complexity
    insert) are rebuilt by the global-atomics kernels inside the same run.  Directory and entries must equal the
    host model's for every genome, and the pair results the oracle's."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    st = SG.Stream(5)
    rnd = lambda n: (st.u64(n) % np.uint64(4)).astype(np.uint8)
    base = rnd(40000)
    seqs = [base, SG.mutate(base, 0.05, st), rnd(36000),
            np.concatenate([rnd(9000), np.tile(np.array([0, 1, 2, 3, 3, 1], np.uint8), 80), rnd(9000)]),
            np.zeros(50000, np.uint8), np.tile(np.array([0, 1, 2, 3, 3, 1], np.uint8), 8000),
            np.concatenate([rnd(8000), np.full(30000, 3, np.uint8), rnd(8000)])]
    lib = U.model_lib()
    eng = L.Engine()
    eng.set_genomes(seqs)
    maxlen = max(len(s) for s in seqs)
    for gid, s in enumerate(seqs):
        d = eng.debug_index(gid)
        T = 2 * len(s) + 3 * eng.params["mrd"]
        wn = (T + 63) // 64 + 2
        t2 = np.zeros(2 * wn, np.uint64); nm = np.zeros(wn, np.uint64)
        dirz = np.zeros(len(d["dirz"]), np.uint32); ent = np.zeros(T + 1, np.uint32)
        n_ent = C.c_uint32(0)
        s = np.ascontiguousarray(s)
        assert lib.model_index(O._ptr(s), len(s), maxlen, O.params_array(None), O._ptr(t2), O._ptr(nm),
                               O._ptr(dirz), O._ptr(ent), C.byref(n_ent)) == 0
        assert np.array_equal(d["dirz"], dirz), gid
        assert n_ent.value == len(d["ent"]) and np.array_equal(_bucketwise_sorted(d["dirz"], d["ent"]), ent[:n_ent.value]), gid
        small = np.diff(d["dirz"].astype(np.int64)) <= 32                     # small buckets come sorted from the device
        for b in np.nonzero(small & (np.diff(d["dirz"].astype(np.int64)) > 1))[0][:2000]:
            seg = d["ent"][d["dirz"][b]:d["dirz"][b + 1]]
            assert np.all(seg[:-1] <= seg[1:]), (gid, int(b))
    eng.close()
    # pairs: without the three degenerate genomes (a poly-A pair is quadratic for every implementation)
    assert np.array_equal(gpu_all2all(seqs[:4]), O.oracle_all2all(seqs[:4], None, threads=16))


@pytest.mark.parametrize("env", [{}, {"LZANI_SORT_INDEX_MIN_DIRBITS": "0"}], ids=["lds-build", "sort-build"])
def test_index_build_mid_size_directories(monkeypatch, env):
    """k_idx_build beyond viral size: 2^18 and 2^19 buckets (16 and 32 bucket ranges per reference); and the sort-based
    build (keys -> radix sort -> one streaming pass), which takes over from 2^20 buckets on, forced at these sizes."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    st = SG.Stream(8)
    a = (st.u64(100_000) % np.uint64(4)).astype(np.uint8)
    b = (st.u64(230_000) % np.uint64(4)).astype(np.uint8)
    sets = ([a, SG.mutate(a, 0.04, st), SG.mutate(a, 0.12, st)], [b, SG.mutate(b, 0.06, st), a])
    lib = U.model_lib()
    for seqs in sets:
        eng = L.Engine()
        eng.set_genomes(seqs)
        maxlen = max(len(s) for s in seqs)
        for gid, s in enumerate(seqs):
            d = eng.debug_index(gid)
            assert d["geom"][1] in (18, 19)
            T = 2 * len(s) + 3 * eng.params["mrd"]
            wn = (T + 63) // 64 + 2
            t2 = np.zeros(2 * wn, np.uint64); nm = np.zeros(wn, np.uint64)
            dirz = np.zeros(len(d["dirz"]), np.uint32); ent = np.zeros(T + 1, np.uint32)
            n_ent = C.c_uint32(0)
            s = np.ascontiguousarray(s)
            assert lib.model_index(O._ptr(s), len(s), maxlen, O.params_array(None), O._ptr(t2), O._ptr(nm),
                                   O._ptr(dirz), O._ptr(ent), C.byref(n_ent)) == 0
            assert np.array_equal(d["dirz"], dirz), gid
            assert n_ent.value == len(d["ent"]) and np.array_equal(_bucketwise_sorted(d["dirz"], d["ent"]), ent[:n_ent.value]), gid
        got = eng.all2all()
        eng.close()
        assert np.array_equal(got, O.oracle_all2all(seqs, None, threads=16))


def test_multi_batch_path(monkeypatch):
    """More reference rows than index slabs (LZANI_MAX_SLOTS=3 forces it; at full size it is the 5 Mbp and the
    100k-genome configurations that batch): dense, ragged/sparse and alignment runs over several batches equal the
    oracle and the single-batch run."""
    _, seqs = SG.make_set(26, 12, lmin=3000, lmax=6000, fam=5)
    seqs[4] = np.concatenate([seqs[4][:700], np.full(25, 5, np.uint8), seqs[4][700:]])
    want = O.oracle_all2all(seqs, None, threads=16)
    n = len(seqs)
    eng = L.Engine()
    eng.set_genomes(seqs)
    single = eng.all2all()
    assert eng.layout()["batches_last_run"] == 1
    ref_ids, row_off = L.dense_rows(n)
    one_regs = eng.run_rows_regions(ref_ids, row_off, None)[1]
    eng.close()
    assert np.array_equal(single, want)

    monkeypatch.setenv("LZANI_MAX_SLOTS", "3")
    eng = L.Engine()
    eng.set_genomes(seqs)
    got = eng.all2all()
    lay = eng.layout()
    assert lay["slots"] == 3 and lay["batches_last_run"] == 9
    assert np.array_equal(got, want)
    # ragged rows: empty rows, repeated references, rows that straddle batch boundaries
    st = SG.Stream(31)
    rr, off, q = [], [0], []
    for k in range(23):
        r = st.randint(0, n - 1)
        cnt = 0 if k % 6 == 1 else st.randint(1, 15)
        rr.append(r)
        for _ in range(cnt):
            x = st.randint(0, n - 1)
            q.append(x if x != r else (x + 1) % n)
        off.append(len(q))
    flat = eng.run_rows(rr, off, q)
    assert eng.layout()["batches_last_run"] == 8
    e = 0
    for row, r in enumerate(rr):
        for x in q[off[row]:off[row + 1]]:
            assert flat[e].tolist() == want[r, x].tolist(), (row, r, x)
            e += 1
    # the alignment instantiation across batches: same regions as the single-batch run
    out, regs = eng.run_rows_regions(ref_ids, row_off, None)
    assert eng.layout()["batches_last_run"] == 9
    eng.close()
    assert np.array_equal(out.reshape(-1, 3), want[~np.eye(n, dtype=bool)])
    assert len(regs) == len(one_regs) and np.array_equal(regs, one_regs)


def test_more_rows_than_grid_limit():
    """70,000 reference rows in one call: beyond the 65,535 slabs one batch can have (the hard limit BASELINE
    configs[4] with its 100,000 genomes runs into) -- sparse one-query rows over tiny genomes, against the oracle."""
    st = SG.Stream(606)
    base = [(st.u64(st.randint(90, 160)) % np.uint64(4)).astype(np.uint8) for _ in range(500)]
    for k in range(0, 500, 5):                                   # make some pairs related so that the results are not all zero
        base[k + 1] = SG.mutate(base[k], 0.03, st)
    n_rows = 70_000
    ref_ids = np.array([st.randint(0, 499) for _ in range(n_rows)], dtype=np.uint32)
    q = np.array([(int(r) // 5 * 5 + 1) if int(r) % 5 == 0 and k % 3 == 0 else (int(r) + 1 + k % 498) % 500 for k, r in enumerate(ref_ids)], dtype=np.uint32)
    assert (q != ref_ids).all()
    row_off = np.arange(n_rows + 1, dtype=np.uint64)
    eng = L.Engine()
    eng.set_genomes(base)
    got = eng.run_rows(ref_ids, row_off, q)
    lay = eng.layout()
    eng.close()
    assert lay["slots"] == 65535 and lay["batches_last_run"] == 2
    full = O.oracle_all2all(base, None, threads=16)
    assert np.array_equal(got, full[ref_ids, q])
    assert (got[:, 0] > 0).sum() > 1000


def test_full_size_10k_genomes():
    """BASELINE configs[2] at full size on one GPU (10,000 x ~40 kbp, 99,990,000 directed pairs -- the
    configuration the headline metric is quoted on): size-independent properties over every pair and exact
    agreement with the oracle on 600 seeded pairs."""
    n = 10_000
    names, seqs = SG.make_set_cached(n, 2)
    lens = np.array([len(s) for s in seqs])
    eng = L.Engine()
    eng.set_genomes(seqs)
    ref_ids, row_off = L.dense_rows(n)
    flat = eng.run_rows(ref_ids, row_off, None).reshape(n, n - 1, 3)
    lay = eng.layout()
    eng.close()
    # dense rows take their candidates from per-pair bitmaps (presence matrix of the batch's references): a batch is
    # bounded by the bitmaps of its pairs -- 64 GB = 1,024 rows of 9,999 pairs -- and every batch is one bitmap-fed launch
    assert lay["tag_words"] == 1 and lay["n_free"] == 1
    assert lay["bitmap_launches"] == lay["batches_last_run"] and 2 <= lay["batches_last_run"] <= 40 and lay["block_launches"] == 0
    mat, lit, comp = flat[..., 0], flat[..., 1], flat[..., 2]
    assert flat.min() >= 0
    assert np.array_equal(comp == 0, (mat == 0) & (lit == 0))
    assert (mat + lit >= 35 * comp).all()                         # a region needs >= reg symbols ...
    qlen = np.empty((n, n - 1), dtype=np.int64)                  # ... and cannot cover more than the query text
    for r in range(n):
        qlen[r, :r] = lens[:r]
        qlen[r, r:] = lens[r + 1:]
    assert (mat + lit <= qlen + 40).all()
    # family structure: every member aligns over most of its length against its family's ancestor (member 0)
    idx = np.arange(n)
    fam0 = idx // 10 * 10
    rel = idx != fam0
    col = np.where(idx < fam0, idx, idx - 1)                      # column of query idx in row fam0 (idx > fam0 always here)
    assert (mat[fam0[rel], col[rel]] > 0.5 * lens[idx[rel]]).all()
    # unrelated random genomes share only chance matches
    other = (idx + 5000) % n
    ocol = np.where(idx < other, idx, idx - 1)
    assert (mat[other, ocol] < 0.05 * lens).all()
    # checksum of checksums: reproducible across runs of the same set (recorded in DESIGN.md)
    print("10k checksum", int(mat.sum(dtype=np.int64)), int(lit.sum(dtype=np.int64)), int(comp.sum(dtype=np.int64)))
    st = SG.Stream(2024)
    pairs = [(st.randint(0, n - 1), st.randint(0, n - 1)) for _ in range(400)]
    pairs += [(10 * (k * 37 % 1000), 10 * (k * 37 % 1000) + 1 + k % 9) for k in range(200)]          # related pairs
    for r, q in pairs:
        if r != q:
            assert tuple(flat[r, q if q < r else q - 1]) == O.oracle_pair(seqs[r], seqs[q]), (r, q)


def test_bacterial_geometry_5mbp(monkeypatch):
    """BASELINE configs[3] at its own geometry: 5 Mbp genomes with --mal 15 --msl 9 --reg 60 -> 30 key bits, 2^24
    buckets, 24-bit positions, 6 tag bits, i.e. tag words + a 256 MB bucket table per slab and the
    global-atomics index build; the whole 5 x 5 matrix against the oracle."""
    st = SG.Stream(303)
    anc = (st.u64(5_000_000) % np.uint64(4)).astype(np.uint8)
    other = (st.u64(4_700_000) % np.uint64(4)).astype(np.uint8)
    seqs = [anc, SG.mutate(anc, 0.01, st), SG.mutate(anc, 0.06, st), other, SG.mutate(other, 0.03, st)]
    prm = dict(mal=15, msl=9, reg=60)
    eng = L.Engine(prm)
    eng.set_genomes(seqs)
    lay = eng.layout()
    assert (lay["key_bits"], lay["dir_bits"], lay["pos_bits"], lay["tag_mask"]) == (30, 24, 24, 0x3F)
    assert lay["kmer_words"] == 1 and lay["bucket_table"] == 1 and lay["tag_words"] == 1 and lay["join_lists"] == 1
    monkeypatch.setenv("LZANI_PM_MIN_ROWS", "32")
    got = eng.all2all()
    assert eng.layout()["bytes_per_slot"] > 400 << 20 and eng.layout()["bitmap_launches"] == 0      # (candidates by the join: the form of filtered rows)
    monkeypatch.delenv("LZANI_PM_MIN_ROWS")
    # the same matrix the way dense rows of long genomes run by themselves (from two rows on since round 4): the candidates
    # from the presence matrix -- 2^30 rows of 16 bytes for a group of up to 128 references, made from the batch's indexes,
    # the pair's bitmap 690 KB -- and, the pairs being this few, every pair by 16 waves (lzani_kernels_split.h)
    got_pm = eng.all2all()
    lay = eng.layout()
    assert lay["bitmap_launches"] == 1 and lay["matrix_from_index"] == 1 and lay["split_launches"] == 1 and lay["split_segments"] >= 16 * 20, lay
    eng.close()
    want = O.oracle_all2all(seqs, prm, threads=16)
    assert np.array_equal(got_pm, want)
    bad = np.argwhere((got != want).any(axis=2))
    assert len(bad) == 0, (bad[:4].tolist(), got[tuple(bad[0])], want[tuple(bad[0])])
    assert got[0, 1, 0] > 4_500_000 and got[3, 4, 0] > 4_000_000 and got[0, 3, 0] < 1_000_000


def test_radix_sort_segments_against_numpy():
    """The engine's own radix sort (lzani_sort.hip) against numpy's stable sort: one segment and many, lengths that are not
    multiples of a tile, one to eight passes, skewed digits (the all-ones filler of positions without a k-mer), ties kept
    in input order (stability: equal sort bits, different low bits)."""
    rng = np.random.default_rng(99)
    eng = L.Engine()
    cases = [(1, 1, 0, 64), (5, 3, 0, 8), (8192, 1, 0, 31), (8193, 2, 17, 48), (100_000, 3, 24, 55), (20_001, 7, 0, 64),
             (1_000_003, 2, 24, 55), (70_000, 1, 32, 56), (3000, 40, 20, 51)]
    for seg_len, n_seg, b0, b1 in cases:
        keys = rng.integers(0, 1 << 63, size=seg_len * n_seg, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=seg_len * n_seg, dtype=np.uint64)
        if seg_len > 1000:
            keys[rng.random(keys.size) < 0.3] = np.uint64(0xFFFFFFFFFFFFFFFF)            # the filler: one digit holds a third of the keys
            keys[: seg_len // 2] &= np.uint64((1 << 40) - 1) | np.uint64(0xFFFF000000000000)    # many ties in the middle bits
        got = eng.debug_sort_segments(keys, seg_len, n_seg, b0, b1)
        mask = np.uint64(((1 << (b1 - b0)) - 1))
        for s in range(n_seg):
            seg = keys[s * seg_len:(s + 1) * seg_len]
            order = np.argsort((seg >> np.uint64(b0)) & mask, kind="stable")
            assert np.array_equal(got[s * seg_len:(s + 1) * seg_len], seg[order]), (seg_len, n_seg, b0, b1, s)
    eng.close()


def test_matrix_from_index_and_longest_pair_first_forced(monkeypatch):
    """The two round-4 pieces of the candidate stage at sizes the oracle covers whole: the presence matrix made from the
    batch's anchor indexes (k_pm_from_index: every row width, several chunks per slot, groups of less than 128 / 512 slots)
    and the ticket order of the queues, longest pair first (k_pm_cand's counts -> k_lpt_keys -> sort).  Both forced by the
    environment; results must equal the default path's and the oracle's."""
    cases = [(SG.make_set(40, 21, lmin=5000, lmax=9000, fam=5)[1], None),
             (SG.make_set(150, 22, lmin=2000, lmax=3000, fam=6)[1], None),                 # rw = 8: two words of slots
             (SG.make_set(24, 23, lmin=60_000, lmax=90_000, fam=4, dmin=0.005, dmax=0.08)[1], dict(mal=13, msl=8, reg=50))]
    for seqs, prm in cases:
        want = O.oracle_all2all(seqs, prm, threads=16)
        eng = L.Engine(prm)
        eng.set_genomes(seqs)
        monkeypatch.setenv("LZANI_PM_MIN_ROWS", "1")
        monkeypatch.setenv("LZANI_PM_FROM_INDEX", "0")
        base = eng.all2all()
        assert eng.layout()["bitmap_launches"] == 1 and eng.layout()["matrix_from_index"] == 0 and eng.layout()["lpt_launches"] == 0
        monkeypatch.setenv("LZANI_PM_FROM_INDEX", "1")
        monkeypatch.setenv("LZANI_LPT", "1")
        got = eng.all2all()
        lay = eng.layout()
        for k in ("LZANI_PM_MIN_ROWS", "LZANI_PM_FROM_INDEX", "LZANI_LPT"):
            monkeypatch.delenv(k)
        eng.close()
        assert lay["matrix_from_index"] >= 1 and lay["lpt_launches"] == 1, lay
        assert np.array_equal(base, want) and np.array_equal(got, want), (len(seqs), prm)


def test_pairs_split_over_several_waves_forced(monkeypatch):
    """Few, long pairs by several waves each (lzani_kernels_split.h: checkpoints, segments, stitch, void segments run again)
    at sizes the oracle covers whole, forced by the environment with cuts every 1,500-6,000 query positions: related and
    unrelated pairs, default and long-k-mer parameters (the hand-written loops inside the segments), genomes with N (the
    generic instantiation's bounds), a run-time parameter tuple on the generic kernel.  Equal to the unsplit run and to
    the oracle; the layout info says the pairs were split."""
    withn = [s.copy() for s in SG.make_set(20, 24, lmin=17000, lmax=21000, fam=5)[1]]
    for k in range(0, 20, 3):
        withn[k][400:400 + 9 + 2 * k] = 5
    cases = [(SG.make_set(30, 21, lmin=17000, lmax=21000, fam=6)[1], None, 2000),
             (SG.make_set(24, 23, lmin=60_000, lmax=90_000, fam=4, dmin=0.005, dmax=0.08)[1], dict(mal=15, msl=9, reg=60), 6000),
             (withn, None, 1500),
             (SG.make_set(16, 25, lmin=30_000, lmax=40_000, fam=4)[1], dict(reg=50, aw=12, am=5), 3000)]
    monkeypatch.setenv("LZANI_PM_MIN_ROWS", "1")
    for seqs, prm, seglen in cases:
        want = O.oracle_all2all(seqs, prm, threads=16)
        eng = L.Engine(prm)
        eng.set_genomes(seqs)
        base = eng.all2all()
        assert eng.layout()["split_launches"] == 0
        monkeypatch.setenv("LZANI_SPLIT", "1")
        monkeypatch.setenv("LZANI_SPLIT_SEGLEN", str(seglen))
        cut_all = seglen != 3000                      # (by default only the pairs with many anchor candidates -- the related ones -- are cut)
        monkeypatch.setenv("LZANI_SPLIT_ALL", "1" if cut_all else "0")
        got = eng.all2all()
        lay = eng.layout()
        for k in ("LZANI_SPLIT", "LZANI_SPLIT_SEGLEN", "LZANI_SPLIT_ALL"):
            monkeypatch.delenv(k)
        eng.close()
        n = len(seqs)
        assert lay["split_launches"] == 1 and lay["split_segments"] >= (2 if cut_all else 1) * n * (n - 1), lay
        print(f"split: {n} genomes, cuts every {seglen}: {lay['split_segments']} segments run for {n * (n - 1)} pairs")
        bad = np.argwhere((got != want).any(axis=2))
        assert len(bad) == 0, (prm, seglen, bad[:4].tolist(), got[tuple(bad[0])], want[tuple(bad[0])])
        assert np.array_equal(base, want)


def test_long_genomes_presence_matrix_natural_trigger():
    """BASELINE configs[3] at its own geometry with enough genomes that the candidate-bitmap form is chosen BY ITSELF (dense
    rows, >= 32 of them: no LZANI_PM_MIN_ROWS): 33 genomes of 4.6-5.2 Mbp in four families, --mal 15 --msl 9 --reg 60 --
    2^30 matrix rows of 16 bytes, 690 KB of bitmap per pair, the null chain of the long-genome parameter set.  48 seeded
    pairs (half of them related) against the reference's parser, size-independent properties on all 1,056."""
    st = SG.Stream(3303)
    prm = dict(mal=15, msl=9, reg=60)
    seqs, fam = [], []
    for f, members in enumerate((9, 9, 8, 7)):
        anc = (st.u64(st.randint(4_600_000, 5_200_000)) % np.uint64(4)).astype(np.uint8)
        seqs.append(anc); fam.append(f)
        for _ in range(members - 1):
            seqs.append(SG.mutate(anc, 0.005 + 0.075 * st.one(), st)); fam.append(f)
    n = len(seqs)
    assert n == 33
    eng = L.Engine(prm)
    eng.set_genomes(seqs)
    got = eng.all2all()
    lay = eng.layout()
    tm = eng.timing()
    eng.close()
    assert lay["bitmap_launches"] >= 1 and lay["bitmap_launches"] == lay["batches_last_run"] and lay["join_lists"] == 1, lay
    # (round 4) at this geometry the matrix comes from the batch's indexes and the tickets go longest pair first, by themselves
    # ... or, with as few pairs as here (1,056 on 8,192 wave slots), every pair is scanned by several waves (lzani_kernels_split.h)
    assert lay["matrix_from_index"] >= 1 and lay["lpt_launches"] + lay["split_launches"] == lay["bitmap_launches"], lay
    print(f"33 x 5 Mbp: pair kernel {tm['pairs_ms']:.0f} ms, candidate stage {tm['cand_ms']:.0f} ms, index {tm['index_ms']:.0f} ms, k-mer words {tm['kmers_ms']:.0f} ms")
    lens = np.array([len(s) for s in seqs])
    fam = np.array(fam)
    for r in range(n):
        for q in range(n):
            mat, lit, comp = (int(x) for x in got[r, q])
            if r == q:
                assert (mat, lit, comp) == (0, 0, 0)
            elif fam[r] == fam[q]:
                assert mat > 0.6 * min(lens[r], lens[q]) and comp >= 1 and mat + lit <= lens[q], (r, q, mat, lit, comp)
            else:
                assert mat + lit < 0.02 * lens[q], (r, q, mat, lit, comp)
    pairs = []
    while len(pairs) < 48:
        r, q = st.randint(0, n - 1), st.randint(0, n - 1)
        if r != q and (fam[r] == fam[q]) == (len(pairs) % 2 == 0):
            pairs.append((r, q))
    rr = np.array([p[0] for p in pairs], np.uint32)
    qq = np.array([p[1] for p in pairs], np.uint32)
    off = np.arange(len(pairs) + 1, dtype=np.uint64)
    if O.lib_ref() is not None:
        want = O.ref_rows(seqs, rr, off, qq, prm, threads=16)
    else:
        want = np.array([O.oracle_pair(seqs[r], seqs[q], prm) for r, q in pairs], dtype=np.int32)
    for k, (r, q) in enumerate(pairs):
        assert tuple(got[r, q]) == tuple(int(x) for x in want[k]), (r, q, got[r, q], want[k])


def test_join_form_of_candidate_detection(monkeypatch):
    """Long genomes find their candidates by a join of the query's sorted k-mer list with the reference's tag words
    instead of one probe per query position.  LZANI_JOIN_MIN_BYTES=1 turns the join on at every size, so the committed
    reference vectors, a seeded fuzz (N runs, inversions, random parameters), a multi-batch run and the filtered row
    form all go through it; results must not change."""
    monkeypatch.setenv("LZANI_JOIN_MIN_BYTES", "1")
    monkeypatch.setenv("LZANI_PM", "0")          # (dense rows of long genomes take candidate bitmaps by themselves: here the join is the subject)
    with open(os.path.join(U.GOLD, "ref_vectors.json")) as f:
        vec = json.load(f)
    done = 0
    for key, item in vec["sets"].items():
        if not key.startswith("synth24"):
            continue
        eng = L.Engine(item["params"])
        eng.set_genomes(_inputs("synth24"))
        join = eng.layout()["join_lists"]
        got = eng.all2all()
        eng.close()
        assert np.array_equal(got, np.array(item["res"], dtype=np.int32)), key
        done += join
    assert done >= 8                                   # the join really ran (where tag words and the anchor queue apply)
    st = SG.Stream(9001)
    for it in range(40):
        prm, seqs = U.fuzz_case_medium(st)
        assert np.array_equal(gpu_all2all(seqs, prm), O.oracle_all2all(seqs, prm, threads=16)), (it, prm)
    monkeypatch.setenv("LZANI_MAX_SLOTS", "4")
    _, seqs = SG.make_set(14, 21, lmin=9000, lmax=16000, fam=7)
    seqs[5] = np.concatenate([seqs[5][:3000], np.full(33, 5, np.uint8), seqs[5][3000:]])
    want = O.oracle_all2all(seqs, None, threads=16)
    eng = L.Engine()
    eng.set_genomes(seqs)
    assert eng.layout()["join_lists"] == 1
    assert np.array_equal(eng.all2all(), want) and eng.layout()["batches_last_run"] == 4
    rr, off, q = [3, 3, 9, 0], [0, 5, 5, 11, 13], [1, 2, 4, 5, 13, 0, 1, 2, 3, 4, 5, 7, 9]
    flat = eng.run_rows(rr, off, q)
    eng.close()
    e = 0
    for row, r in enumerate(rr):
        for x in q[off[row]:off[row + 1]]:
            assert flat[e].tolist() == want[r, x].tolist(), (row, r, x)
            e += 1


def test_fuzz_medium_block_and_bitmap_kernels(monkeypatch):
    """The differential fuzz of tools/fuzz_gpu.py at 8-70 kbp (random parameters in three cases out of four, N runs,
    inversions), 20 seeded cases, each through three forms of the pair path: the default one (4 dense rows: the wave
    kernel with a probe per position), rows of 135 pairs through the block kernel with the LDS filter, and dense rows
    with their candidates from the presence matrix (forced from one row on)."""
    st = SG.Stream(20260)
    blk = pmx = 0
    for case in range(20):
        prm, seqs = U.fuzz_case_medium(st)
        want = O.oracle_all2all(seqs, prm, threads=16)
        eng = L.Engine(prm)
        eng.set_genomes(seqs)
        assert np.array_equal(eng.all2all(), want), (case, prm)
        monkeypatch.setenv("LZANI_PM_MIN_ROWS", "1")
        got = eng.all2all()
        pmx += eng.layout()["bitmap_launches"]
        monkeypatch.delenv("LZANI_PM_MIN_ROWS")
        assert np.array_equal(got, want), ("bitmaps", case, prm)
        n = len(seqs)
        qs = [[x for x in range(n) if x != r] * 45 for r in range(n)]
        row_off = np.zeros(n + 1, np.uint64)
        row_off[1:] = np.cumsum([len(x) for x in qs])
        monkeypatch.setenv("LZANI_BLOCK_KERNEL", "1")
        out = eng.run_rows(np.arange(n, dtype=np.uint32), row_off, np.array([x for row in qs for x in row], np.uint32)).reshape(-1, 3)
        blk += eng.layout()["block_launches"]
        monkeypatch.delenv("LZANI_BLOCK_KERNEL")
        eng.close()
        assert np.array_equal(out, np.concatenate([want[r, qs[r]] for r in range(n)])), ("block kernel", case, prm)
    assert blk >= 10 and pmx >= 10, (blk, pmx)       # (both need tag words: not every random parameter set has them)


def test_null_chain_long_genome_parameters():
    """The hand-written null chain is compiled for two parameter sets: the defaults and --mal 15 --msl 9 --reg 60
    (BASELINE configs[3]; the seed bitmap then works on the 14 hash bits k_kmers packs above the 9-mer).  Viral-size
    and mid-size genomes with the second set through the bitmap form -- unrelated pairs (runs of null events), close
    families (seed events, kept regions), N runs and ragged lengths -- against the oracle; the same data with msl 8
    (packed k-mer words, generic kernel)."""
    prm = dict(mal=15, msl=9, reg=60)
    _, viral = SG.make_set(60, 91, lmin=30000, lmax=44000, fam=6, dmin=0.01, dmax=0.12)
    withn = [s.copy() for s in viral[:40]]
    for k in range(0, 40, 5):
        withn[k][500 + 37 * k:500 + 37 * k + 3 + k] = 5
    withn = [s[: 4000 + (k * 977) % (len(s) - 4000)] for k, s in enumerate(withn)]
    _, mid = SG.make_set(12, 17, lmin=250_000, lmax=300_000, fam=3, dmin=0.005, dmax=0.05)
    for name, data, p in (("viral", viral, prm), ("N runs, ragged", withn, prm), ("mid-size", mid, prm), ("msl 8", viral[:40], dict(mal=12, msl=8))):
        eng = L.Engine(p)
        eng.set_genomes(data)
        got = eng.all2all()
        lay = eng.layout()
        eng.close()
        assert lay["bitmap_launches"] == 1, (name, lay)
        want = O.oracle_all2all(data, p, threads=16)
        bad = np.argwhere((got != want).any(axis=2))
        assert len(bad) == 0, (name, bad[:3].tolist(), got[tuple(bad[0])], want[tuple(bad[0])])


def test_presence_matrix_candidates(monkeypatch):
    """Dense rows: the candidates of every pair come from per-pair bitmaps made ahead from the presence matrix of the
    batch's references (k_pm_build, k_pm_cand) and the pair kernel reads them 64 positions per lane: whole matrices
    against the oracle -- N-free and with N (poly-N stretches, an all-N genome), default and other parameters (mal 9
    and 12: 18 and 24 key bits), genomes of very different lengths (tiles of 1,024 positions: queries of 1 to 20 tiles),
    more rows than one group of 512 references and more than one batch (LZANI_PM_MAX_BYTES bounds the bitmaps)."""
    _, seqs = SG.make_set(150, 31, lmin=17000, lmax=20000, fam=10)      # (tag words need >= 2^15 buckets)
    withn = [s.copy() for s in seqs]
    for k in range(0, 150, 7):
        withn[k][300:300 + 5 + k % 40] = 5
    withn.append(np.full(700, 5, dtype=np.uint8))
    ragged = [s[: 900 + (k * 811) % len(s)] for k, s in enumerate(seqs[:60])]
    _, longer = SG.make_set(40, 12, lmin=34000, lmax=36000, fam=8)      # (mal 12: 24 key bits, 2^17 buckets, 7 tag bits)
    for name, data, prms in (("N-free", seqs, (None, dict(reg=30, aw=20), dict(mal=9, msl=6))),
                             ("with N", withn, (None,)), ("ragged", ragged, (None,)), ("mal 12", longer, (dict(mal=12),))):
        for prm in prms:
            eng = L.Engine(prm)
            eng.set_genomes(data)
            got = eng.all2all()
            lay = eng.layout()
            eng.close()
            assert lay["bitmap_launches"] == 1 and lay["block_launches"] == 0 and lay["tag_words"] == 1, (name, prm, lay)
            want = O.oracle_all2all(data, prm, threads=16)
            bad = np.argwhere((got != want).any(axis=2))
            assert len(bad) == 0, (name, prm, bad[:3].tolist())
    # several groups of 512 references in one batch, then several batches (the bitmaps of 64 rows at a time)
    _, small = SG.make_set(1100, 77, lmin=1500, lmax=2500, fam=10)
    prm = dict(mal=9, msl=6)
    want = O.oracle_all2all(small, prm, threads=16)
    eng = L.Engine(prm)
    eng.set_genomes(small)
    got = eng.all2all()
    lay = eng.layout()
    assert lay["bitmap_launches"] == 1 and lay["batches_last_run"] == 1, lay
    assert np.array_equal(got, want)
    monkeypatch.setenv("LZANI_PM_MAX_BYTES", str(64 * 1099 * 3 * 128))     # 64 rows of 1,099 bitmaps of 3 tiles
    got = eng.all2all()
    lay = eng.layout()
    eng.close()
    assert lay["bitmap_launches"] == lay["batches_last_run"] == 18, lay
    assert np.array_equal(got, want)


def test_presence_matrix_after_sparse_run_and_allocation_fallback(monkeypatch):
    """One Engine, a sparse run with many rows first (its index slabs grow to one per row), then a dense run: the
    bitmap form counts the context's slabs as available, so slabs larger than it wants are released before it sizes its
    buffers.  And a failed allocation of the candidate bitmaps is not an error: the run falls back to the
    probe form (LZANI_PM_FAIL_CBITS=1 makes the request unsatisfiable)."""
    _, seqs = SG.make_set(150, 47, lmin=17000, lmax=20000, fam=10)
    n = len(seqs)
    want = O.oracle_all2all(seqs, None, threads=16)
    eng = L.Engine()
    eng.set_genomes(seqs)
    # sparse rows: every genome as a reference with two queries each -> 150 slabs, probe form
    rr = np.arange(n, dtype=np.uint32)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(2)
    q = np.array([[(r + 1) % n, (r + 77) % n] for r in range(n)], dtype=np.uint32).reshape(-1)
    got = eng.run_rows(rr, off, q)
    lay = eng.layout()
    assert lay["bitmap_launches"] == 0 and lay["slots"] == n, lay
    for r in range(n):
        assert tuple(got[2 * r]) == tuple(want[r, (r + 1) % n]) and tuple(got[2 * r + 1]) == tuple(want[r, (r + 77) % n])
    # dense rows of a few references on the same context: bitmap form, slabs shrunk to what it wants
    monkeypatch.setenv("LZANI_PM_MIN_ROWS", "1")
    ref_ids, row_off = L.dense_rows(n, np.arange(40, dtype=np.uint32))
    got = eng.run_rows(ref_ids, row_off, None).reshape(40, n - 1, 3)
    lay = eng.layout()
    assert lay["bitmap_launches"] == 1 and lay["slots"] == 40, lay
    for r in range(40):
        assert np.array_equal(got[r], want[r, np.arange(n) != r]), r
    # the same rows with the bitmaps' allocation failing: fallback, same results
    monkeypatch.setenv("LZANI_PM_FAIL_CBITS", "1")
    got = eng.run_rows(ref_ids, row_off, None).reshape(40, n - 1, 3)
    lay = eng.layout()
    assert lay["bitmap_launches"] == 0, lay
    monkeypatch.delenv("LZANI_PM_FAIL_CBITS")
    for r in range(40):
        assert np.array_equal(got[r], want[r, np.arange(n) != r]), r
    # and back again
    assert np.array_equal(eng.all2all(), want)
    assert eng.layout()["bitmap_launches"] == 1
    eng.close()


def test_run_time_compiled_parameter_tuples(monkeypatch, tmp_path):
    """Every parameter tuple other than the two compiled ahead of time gets the pair kernel compiled FOR it when the context
    first needs it (lzani_rtc.h: the eight ints folded in, the hand-written null chain included where the tuple is inside
    its envelope): tuples inside and outside the envelope, msl 8 (its own seed-bitmap mapping), dense rows (candidate
    bitmaps) and filtered rows (probes), genomes with and without N -- every launch a run-time compiled one, results
    equal to the oracle's; the second context finds the code objects in the disk cache."""
    monkeypatch.setenv("LZANI_RTC_MIN_PAIRS", "0")
    monkeypatch.setenv("LZANI_RTC_CACHE", str(tmp_path))
    monkeypatch.setenv("LZANI_PM_MIN_ROWS", "1")
    _, seqs = SG.make_set(30, 53, lmin=17000, lmax=21000, fam=6)
    withn = [s.copy() for s in seqs[:18]]
    for k in range(0, 18, 4):
        withn[k][500:500 + 7 + 3 * k] = 5
    tuples = (dict(reg=36), dict(am=6), dict(mal=12, msl=8, mrd=50, mqd=30, reg=40, aw=12, am=5, ar=2),
              dict(mal=9, msl=5, mrd=64, mqd=60, aw=2, am=0, ar=1), dict(aw=20, reg=50))
    for prm, chain in zip(tuples, (1, 1, 1, 1, 0)):
        for name, data in (("N-free", seqs), ("with N", withn)):
            n = len(data)
            want = O.oracle_all2all(data, prm, threads=16)
            eng = L.Engine(prm)
            eng.set_genomes(data)
            got = eng.all2all()
            lay = eng.layout()
            assert lay["bitmap_launches"] == 1 and lay["rtc_launches"] == 1, (prm, name, lay, eng.rtc_info())
            bad = np.argwhere((got != want).any(axis=2))
            assert len(bad) == 0, (prm, name, "dense", bad[:3].tolist())
            rr = np.arange(n, dtype=np.uint32)
            off = np.arange(n + 1, dtype=np.uint64) * np.uint64(3)
            q = np.array([[(r + 1) % n, (r + 2) % n, (r + 7) % n] for r in range(n)], dtype=np.uint32).reshape(-1)
            out = eng.run_rows(rr, off, q).reshape(n, 3, 3)
            lay = eng.layout()
            # (filtered rows take the anchor queue -- and with it the run-time compiled kernel -- by the probe form where the
            # index has tag words: tags of up to 7 bits; mal 12 at this size has 8, and since round 4 such rows take the
            # candidate bitmaps from two pairs per query on instead of the rounds of the first kernel)
            assert lay["bitmap_launches"] == 1 - lay["tag_words"] and lay["rtc_launches"] == 1, (prm, name, lay)
            for r in range(n):
                assert np.array_equal(out[r], want[r, [(r + 1) % n, (r + 2) % n, (r + 7) % n]]), (prm, name, "rows", r)
            info = eng.rtc_info()
            eng.close()
            assert info["folded_ahead_of_time"] == 0 and info["null_chain"] == chain and info["kernels_built"] == 1 + lay["tag_words"] and info["kernels_failed"] == 0, info
    # the same tuple again: from the disk cache
    eng = L.Engine(tuples[0])
    eng.set_genomes(seqs)
    eng.all2all()
    info = eng.rtc_info()
    eng.close()
    assert info["kernels_built"] == 1 and info["kernels_from_cache"] == 1, info
    # below the threshold nothing is compiled, and LZANI_RTC=0 turns it off
    monkeypatch.setenv("LZANI_RTC_MIN_PAIRS", "100000000")
    eng = L.Engine(dict(reg=37))
    eng.set_genomes(seqs[:8])
    got = eng.all2all()
    assert eng.layout()["rtc_launches"] == 0 and eng.rtc_info()["kernels_built"] == 0
    eng.close()
    assert np.array_equal(got, O.oracle_all2all(seqs[:8], dict(reg=37), threads=16))
    eng = L.Engine()
    assert eng.rtc_info()["folded_ahead_of_time"] == 1
    eng.close()


def test_presence_matrix_with_query_lists():
    """Candidate bitmaps for rows with query LISTS that are dense where they are: the row x column blocks a host cuts
    a dense all2all into so that it can emit finished rows while the GPU works on the next block (lz-ani does) --
    block A = rows A against the queries from A on, plus the rows behind A against the queries of A.  Every pair of
    the matrix comes out exactly once over the blocks and equals the oracle; sparse lists (a kmer-db filter's) and
    lists that name a query twice keep the probe form."""
    _, seqs = SG.make_set(150, 31, lmin=17000, lmax=20000, fam=10)
    n = len(seqs)
    want = O.oracle_all2all(seqs, None, threads=16)
    eng = L.Engine()
    eng.set_genomes(seqs)
    got = np.zeros_like(want)
    seen = np.zeros((n, n), dtype=np.int32)
    bk = 40
    for a0 in range(0, n, bk):
        a1 = min(n, a0 + bk)
        rows, lists = [], []
        for r in range(a0, a1):
            rows.append(r); lists.append([q for q in range(a0, n) if q != r])
        for r in range(a1, n):
            rows.append(r); lists.append(list(range(a0, a1)))
        keep = [k for k in range(len(rows)) if lists[k]]
        rows = [rows[k] for k in keep]; lists = [lists[k] for k in keep]
        off = np.zeros(len(rows) + 1, np.uint64)
        off[1:] = np.cumsum([len(x) for x in lists])
        out = eng.run_rows(np.array(rows, np.uint32), off, np.array([q for x in lists for q in x], np.uint32)).reshape(-1, 3)
        lay = eng.layout()
        assert lay["bitmap_launches"] == lay["batches_last_run"] >= 1 if len(rows) >= 32 else True, (a0, lay)
        e = 0
        for r, x in zip(rows, lists):
            for q in x:
                got[r, q] = out[e]; seen[r, q] += 1; e += 1
    assert (seen + np.eye(n, dtype=np.int32) == 1).all()
    assert np.array_equal(got, want)
    # sparse lists: three relatives per row -> the probe form
    rows = np.arange(n, dtype=np.uint32)
    lists = [[(r + d) % n for d in (1, 2, 3)] for r in range(n)]
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(3)
    out = eng.run_rows(rows, off, np.array([q for x in lists for q in x], np.uint32)).reshape(n, 3, 3)
    assert eng.layout()["bitmap_launches"] == 0
    assert all(np.array_equal(out[r, k], want[r, lists[r][k]]) for r in range(n) for k in range(3))
    # a query twice in a row: one bitmap per (row, query) would not do -> the probe form as well
    lists = [[q for q in range(n) if q != r] + [(r + 1) % n] for r in range(64)]
    off = np.zeros(65, np.uint64)
    off[1:] = np.cumsum([len(x) for x in lists])
    out = eng.run_rows(np.arange(64, dtype=np.uint32), off, np.array([q for x in lists for q in x], np.uint32)).reshape(-1, 3)
    assert eng.layout()["bitmap_launches"] == 0
    eng.close()
    e = 0
    for r, x in enumerate(lists):
        for q in x:
            assert np.array_equal(out[e], want[r, q]), (r, q)
            e += 1


def test_filtered_rows_of_mid_size_genomes_with_long_kmers(monkeypatch):
    """Filtered rows (query lists a kmer-db filter leaves: the relatives and a few chance hits per row) of mid-size genomes
    at --mal 15: the tags of such an index do not fit a tag byte, so there are no tag words and the probe form falls back to
    the rounds of the first kernel.  Since round 4 these rows take the candidate bitmaps from two pairs per query and group
    on (one matrix row read serves them; the bitmap form of refill reads the bucket whole): bit-exact against the oracle,
    and timed against the rounds (24,576 pairs: three per wave slot)."""
    import time
    st = SG.Stream(4404)
    _, seqs = SG.make_set(768, 5, lmin=90_000, lmax=110_000, fam=4, dmin=0.005, dmax=0.08)
    n = len(seqs)
    prm = dict(mal=15, msl=9, reg=60)
    lists = []
    for r in range(n):
        fam = [q for q in range((r // 4) * 4, (r // 4) * 4 + 4) if q != r]
        extra = []
        while len(extra) < 29:
            q = st.randint(0, n - 1)
            if q != r and q not in fam and q not in extra:
                extra.append(q)
        lists.append(sorted(fam + extra))
    rows = np.arange(n, dtype=np.uint32)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum([len(x) for x in lists])
    qq = np.array([q for x in lists for q in x], np.uint32)
    if O.lib_ref() is not None:
        want = O.ref_rows(seqs, rows, off, qq, prm, threads=16)
    else:
        want = np.array([O.oracle_pair(seqs[r], seqs[q], prm) for r, x in enumerate(lists) for q in x], dtype=np.int32)
    eng = L.Engine(prm)
    eng.set_genomes(seqs)
    assert eng.layout()["tag_words"] == 0
    got = eng.run_rows(rows, off, qq).reshape(-1, 3)           # (also makes the k-mer words: not in either timing below)
    lay = eng.layout()
    assert lay["bitmap_launches"] == 1, lay
    assert np.array_equal(got, want.reshape(-1, 3))
    t = time.perf_counter(); eng.run_rows(rows, off, qq); t_pm = time.perf_counter() - t
    tm = eng.timing()
    monkeypatch.setenv("LZANI_PM", "0")
    t = time.perf_counter(); got0 = eng.run_rows(rows, off, qq).reshape(-1, 3); t_rounds = time.perf_counter() - t
    tm0 = eng.timing()
    assert eng.layout()["bitmap_launches"] == 0
    monkeypatch.delenv("LZANI_PM")
    eng.close()
    assert np.array_equal(got0, want.reshape(-1, 3))
    print(f"768 x 100 kbp at mal 15, filtered rows of 32 queries ({len(qq)} pairs): {t_pm * 1e3:.1f} ms with candidate bitmaps "
          f"(index {tm['index_ms']:.1f}, candidate stage {tm['cand_ms']:.1f}, pair kernel {tm['pairs_ms']:.1f}), "
          f"{t_rounds * 1e3:.1f} ms by rounds (index {tm0['index_ms']:.1f}, pair kernel {tm0['pairs_ms']:.1f})")
    assert t_pm * 1.2 < t_rounds, (t_pm, t_rounds)          # (1.9x when measured; a timing assertion gets room)


def test_block_kernel_with_lds_filter(monkeypatch):
    """Rows of >= 128 pairs run by blocks of 16 waves that keep the reference's presence filter in LDS (k_pairs_blk,
    with the null chain where the parameters are the defaults): whole matrices against the oracle -- N-free and with N,
    default and other parameters, 149-pair rows cut across chunks of 128, a filtered CSR with long and short rows --
    and equal to the wave kernel's results; the layout info says which kernel ran."""
    _, seqs = SG.make_set(150, 31, lmin=17000, lmax=20000, fam=10)      # (tag words need >= 2^15 buckets)
    withn = [s.copy() for s in seqs]
    for k in range(0, 150, 7):
        withn[k][300:300 + 5 + k % 40] = 5
    want = {}
    # (dense rows take their candidates from the presence matrix by default -- test_presence_matrix_candidates; the
    # block kernel serves them where that form does not apply: mal > 12, fewer than 32 rows, LZANI_PM=0)
    monkeypatch.setenv("LZANI_PM", "0")
    for name, data, prms in (("N-free", seqs, (None, dict(reg=30, aw=20))), ("with N", withn, (None,))):
        for prm in prms:
            eng = L.Engine(prm)
            eng.set_genomes(data)
            got = eng.all2all()
            lay = eng.layout()
            eng.close()
            assert lay["block_launches"] == 1 and lay["bitmap_launches"] == 0 and lay["tag_words"] == 1, (name, prm)
            want[(name, str(prm))] = O.oracle_all2all(data, prm, threads=16)
            bad = np.argwhere((got != want[(name, str(prm))]).any(axis=2))
            assert len(bad) == 0, (name, prm, bad[:3].tolist())
    # the wave kernel on the same set
    monkeypatch.setenv("LZANI_BLOCK_KERNEL", "0")
    eng = L.Engine()
    eng.set_genomes(seqs)
    got = eng.all2all()
    assert eng.layout()["block_launches"] == 0
    monkeypatch.delenv("LZANI_BLOCK_KERNEL")
    assert np.array_equal(got, want[("N-free", "None")])
    # filtered rows: 35 rows of 140 queries, 34 rows of 1-3 queries in between, an empty row
    st = SG.Stream(5)
    ref_ids, row_off, q = [], [0], []
    for r in range(0, 140, 2):
        if r % 4 == 0:
            qs = [x for x in range(150) if x != r][:140]
        elif r == 70:
            qs = []
        else:
            qs = [x for x in ((r + 1 + st.randint(0, 100)) % 150 for _ in range(st.randint(1, 3))) if x != r]
        ref_ids.append(r); q += qs; row_off.append(len(q))
    ref_ids = np.array(ref_ids, np.uint32); row_off = np.array(row_off, np.uint64); q = np.array(q, np.uint32)
    out = eng.run_rows(ref_ids, row_off, q).reshape(-1, 3)
    assert eng.layout()["block_launches"] == 0                 # filtered rows go to the wave kernel unless told otherwise
    monkeypatch.setenv("LZANI_BLOCK_KERNEL", "1")
    full = want[("N-free", "None")]
    e = 0
    for k, r in enumerate(ref_ids):
        for x in q[int(row_off[k]):int(row_off[k + 1])]:
            assert tuple(out[e]) == tuple(full[r, x]), (k, r, x)
            e += 1
    # the same rows, the long ones only: the block kernel on a CSR
    keep = [k for k in range(len(ref_ids)) if row_off[k + 1] - row_off[k] >= 128]
    r2 = ref_ids[keep]
    q2 = np.concatenate([q[int(row_off[k]):int(row_off[k + 1])] for k in keep])
    off2 = np.zeros(len(keep) + 1, np.uint64)
    off2[1:] = np.cumsum([int(row_off[k + 1] - row_off[k]) for k in keep])
    out2 = eng.run_rows(r2, off2, q2).reshape(-1, 3)
    assert eng.layout()["block_launches"] == 1
    eng.close()
    e = 0
    for k, r in enumerate(r2):
        for x in q2[int(off2[k]):int(off2[k + 1])]:
            assert tuple(out2[e]) == tuple(full[r, x]), (k, r, x)
            e += 1
    assert e == len(q2) and len(keep) >= 30


def test_last_slot_number_is_not_the_invalid_key(monkeypatch):
    """Found by the differential fuzz (tools/fuzz_gpu.py, seed 1202 case 1361) with the join and the sort-based index
    build forced on: both sort 64-bit keys `number || hash || position` whose invalid form is all ones, looking only at
    the number and hash bits -- so the LAST genome / slot of a power-of-two count (here 2) had the all-ones number and
    lost the k-mers of its all-ones hash bucket to the invalid keys.  Two unrelated ~65 kbp genomes at mal = 10."""
    z = np.load(os.path.join(U.GOLD, "fuzz_last_slot_all_ones.npz"))
    prm = dict(zip(("mal", "msl", "mrd", "mqd", "reg", "aw", "am", "ar"), [int(x) for x in z["prm"]]))
    seqs = []
    for name in ("a", "b"):
        pk = z[name]
        s = np.stack([pk & 3, (pk >> 2) & 3, (pk >> 4) & 3, (pk >> 6) & 3], axis=1).reshape(-1)[: int(z[name + "_len"][0])]
        seqs.append(np.ascontiguousarray(s, dtype=np.uint8))
    want = O.oracle_all2all(seqs, prm, threads=4)
    for env in ({}, {"LZANI_JOIN_MIN_BYTES": "1"}, {"LZANI_SORT_INDEX_MIN_DIRBITS": "0"},
                {"LZANI_JOIN_MIN_BYTES": "1", "LZANI_SORT_INDEX_MIN_DIRBITS": "0"}):
        for k in ("LZANI_JOIN_MIN_BYTES", "LZANI_SORT_INDEX_MIN_DIRBITS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        assert np.array_equal(gpu_all2all(seqs, prm), want), env


def test_device_group_single_process(monkeypatch):
    """lzani_group_* (what `lz-ani --gpus n` calls): rows partitioned over the devices of the group (cyclic for dense
    rows, LPT for filtered ones), one context and host thread per device, shards gathered on the first device,
    restored to the caller's CSR order there, one copy out.  On this one-GPU box the group lists device 0 three
    times (a rehearsal: shards move by device copies instead of ncclSend/ncclRecv); results must equal the plain
    single-context run and the oracle."""
    _, seqs = SG.make_set(31, 19, lmin=3000, lmax=7000, fam=6)
    want = O.oracle_all2all(seqs, None, threads=16)
    n = len(seqs)
    for devs in ((0,), (0, 0, 0)):
        grp = L.Group(None, devs)
        grp.set_genomes(seqs)
        ref_ids, row_off = L.dense_rows(n)
        flat = grp.run_rows(ref_ids, row_off, None)
        assert np.array_equal(flat, want[~np.eye(n, dtype=bool)]), devs
        # filtered rows with very unequal sizes, unsorted, some empty
        st = SG.Stream(77)
        rr, off, q = [], [0], []
        for k in range(40):
            r = st.randint(0, n - 1)
            cnt = 0 if k % 7 == 3 else (25 if k % 9 == 0 else st.randint(1, 4))
            rr.append(r)
            for _ in range(cnt):
                x = st.randint(0, n - 1)
                q.append(x if x != r else (x + 1) % n)
            off.append(len(q))
        got = grp.run_rows(rr, off, q)
        e = 0
        for row, r in enumerate(rr):
            for x in q[off[row]:off[row + 1]]:
                assert got[e].tolist() == want[r, x].tolist(), (devs, row, r, x)
                e += 1
        tms = [grp.timing(d) for d in range(len(devs))]
        assert sum(t["pairs"] for t in tms) == len(q)
        if len(devs) > 1:
            assert all(t["pairs"] > 0 for t in tms)
        assert grp.run_rows([], [0], None).shape == (0, 3)
        grp.close()


def _need_two_gpus(why):
    """Skip unless two GPUs are visible.  Asked inside the test, not in a skipif decorator: collecting the module (also
    with the gpu tests deselected, also on a box without torch) must not start the HIP runtime."""
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:
        n = 0
    if n < 2:
        pytest.skip(why)


def test_device_group_two_gpus_rccl():
    """The group on two real devices: shards move by grouped ncclSend / ncclRecv from a single thread across the
    communicators of ncclCommInitAll.  Dense rows (cyclic deal) and filtered rows of very unequal sizes (LPT) against
    the single-context run."""
    _need_two_gpus("needs two GPUs: RCCL over xGMI (the one-GPU box rehearses the same paths on device 0)")
    _, seqs = SG.make_set(41, 29, lmin=3000, lmax=7000, fam=6)
    n = len(seqs)
    eng = L.Engine()
    eng.set_genomes(seqs)
    want = eng.all2all()
    st = SG.Stream(78)
    rr, off, q = [], [0], []
    for k in range(60):
        r = st.randint(0, n - 1)
        cnt = 0 if k % 7 == 3 else (35 if k % 9 == 0 else st.randint(1, 4))
        rr.append(r)
        for _ in range(cnt):
            x = st.randint(0, n - 1)
            q.append(x if x != r else (x + 1) % n)
        off.append(len(q))
    want_rows = eng.run_rows(np.array(rr, dtype=np.uint32), np.array(off, dtype=np.uint64), np.array(q, dtype=np.uint32))
    eng.close()
    grp = L.Group(None, (0, 1))
    grp.set_genomes(seqs)
    ref_ids, row_off = L.dense_rows(n)
    assert np.array_equal(grp.run_rows(ref_ids, row_off, None), want[~np.eye(n, dtype=bool)])
    assert np.array_equal(grp.run_rows(rr, off, q), want_rows)
    assert all(grp.timing(d)["pairs"] > 0 for d in range(2))
    grp.close()


def test_bench_two_ranks_library_collective():
    """bench.py --gpus 2 --collective lzani under torch.distributed.run, one rank per GPU: what the default stays away from
    (--collective torch) until this has passed on hardware."""
    _need_two_gpus("needs two GPUs: the library's own communicator (ncclCommInitRank + ncclAllGather) at N = 2")
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    for coll in ("lzani", "torch"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
               "--genomes", "300", "--seed", "9", "--slab", "50", "--lmin", "3000", "--lmax", "5000", "--collective", coll]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, (coll, p.stderr[-2000:])
        d = json.loads([ln for ln in p.stdout.split("\n") if ln.startswith("{")][0])
        assert d["n_gpus"] == 2 and d["parity_on_last_slab"] == "bit-exact", coll


def test_rccl_communicator_single_rank():
    """lzani_comm_* with a one-rank communicator: the library's own RCCL calls (ncclGetUniqueId, ncclCommInitRank,
    ncclAllGather, grouped ncclSend/ncclRecv path of the root) run on this box; the N > 1 data movement is covered
    on CPU by tests/test_dist.py and measured by the driver's scaling bench."""
    import torch
    _, seqs = SG.make_set(9, 3, lmin=2000, lmax=3000, fam=3)
    eng = L.Engine()
    eng.set_genomes(seqs)
    eng.comm_init(1, 0, L.comm_unique_id())
    n = len(seqs)
    ref_ids, row_off = L.dense_rows(n)
    shard = torch.zeros(n * (n - 1) * 3, dtype=torch.int32, device="cuda")
    recv = torch.full_like(shard, -1)
    eng.run_rows_device(ref_ids, row_off, None, shard.data_ptr())
    eng.comm_allgather(shard.data_ptr(), recv.data_ptr(), n * (n - 1))
    want = O.oracle_all2all(seqs, None, threads=4)[~np.eye(n, dtype=bool)]
    assert np.array_equal(recv.cpu().numpy().reshape(-1, 3), want)
    recv.fill_(-1)
    eng.comm_gatherv(shard.data_ptr(), recv.data_ptr(), [n * (n - 1)], root=0)
    assert np.array_equal(recv.cpu().numpy().reshape(-1, 3), want)
    with pytest.raises(L.LzaniError, match="LZANI_ERR_STATE"):
        eng.comm_init(1, 0, L.comm_unique_id())
    eng.close()


def test_degenerate_inputs():
    """Single genome, empty and sub-k-mer genomes, many tiny genomes: the engine must neither fault nor
    differ from the oracle (which returns zeros for everything too short to hold a seed)."""
    eng = L.Engine()
    one = [SG.make_set(1, 3, lmin=2000, lmax=2100)[1][0]]
    eng.set_genomes(one)
    assert eng.all2all().shape == (1, 1, 3)
    tiny = [np.zeros(0, np.uint8), np.array([0, 1, 2], np.uint8), np.array([3] * 6, np.uint8), one[0][:11].copy(), one[0][:40].copy(), one[0]]
    eng.set_genomes(tiny)
    assert np.array_equal(eng.all2all(), O.oracle_all2all(tiny, None, threads=4))
    st = SG.Stream(8)
    many = [(st.u64(st.randint(60, 140)) % np.uint64(4)).astype(np.uint8) for _ in range(600)]
    many[5] = many[4].copy()
    many[9] = (3 - many[4][::-1]).astype(np.uint8)
    eng.set_genomes(many)
    got = eng.all2all()
    eng.close()
    want = O.oracle_all2all(many, None, threads=16)
    assert np.array_equal(got, want)
    assert got[4, 5, 0] == len(many[4]) and got[4, 9, 0] == len(many[4])


def test_differential_fuzz_gpu_vs_oracle():
    """Seeded random parameters (every instantiation and fallback path gets hit) and sequences through
    the C-ABI against the oracle."""
    st = SG.Stream(4242)
    for it in range(120):
        prm, seqs = U.fuzz_case(st)
        got = gpu_all2all(seqs, prm)
        want = O.oracle_all2all(seqs, prm, threads=4)
        assert np.array_equal(got, want), (it, prm, np.argwhere((got != want).any(axis=2))[:3].tolist())


def test_differential_fuzz_medium_genomes():
    """The same at 8-70 kbp: tag words, bucket table and the LDS index build are in use, parameters random
    (mal <= 12), genomes with N runs and inversions."""
    st = SG.Stream(777)
    for it in range(60):
        prm, seqs = U.fuzz_case_medium(st)
        got = gpu_all2all(seqs, prm)
        want = O.oracle_all2all(seqs, prm, threads=16)
        assert np.array_equal(got, want), (it, prm, np.argwhere((got != want).any(axis=2))[:3].tolist())
