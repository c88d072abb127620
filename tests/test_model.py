"""CPU: the kernel formulation itself (lz-ani_amd/csrc/lzani_core.h: rounds of 64 speculative steps,
mask folds, chunked extensions, O(1) region state), run through the lane-emulating host model,
against the oracle.  This is what lets kernel logic be debugged without a GPU."""
import numpy as np
import pytest

import oracle as O
import synth_genomes as SG
import util as U


@pytest.mark.parametrize("variant", list(U.VARIANTS))
def test_model_example(variant):
    _, seqs = U.load_example()
    prm = U.VARIANTS[variant]
    seqs = seqs[:6] if variant != "default" else seqs
    assert np.array_equal(U.model_all2all(seqs, prm), O.oracle_all2all(seqs, prm, threads=8))


EXTRA = [dict(mrd=0), dict(ar=1), dict(aw=64, am=20), dict(mqd=64, mrd=64), dict(mqd=0), dict(reg=1), dict(am=0),
         dict(mal=7, msl=7), dict(mal=20, msl=12), dict(ar=0), dict(mal=32, msl=16), dict(msl=1, mal=4)]


@pytest.mark.parametrize("prm", list(U.VARIANTS.values()) + EXTRA, ids=str)
def test_model_edge_set(prm):
    seqs = U.edge_set()
    assert np.array_equal(U.model_all2all(seqs, prm), O.oracle_all2all(seqs, prm, threads=8))


def test_model_vir61_subset():
    _, seqs = U.load_vir61()
    seqs = seqs[::5]
    assert np.array_equal(U.model_all2all(seqs), O.oracle_all2all(seqs, None, threads=8))


def test_model_synthetic_families():
    _, seqs = SG.make_set(16, 3, lmin=5000, lmax=9000, fam=4)
    for prm in (None, dict(mal=15, msl=9, reg=60)):
        assert np.array_equal(U.model_all2all(seqs, prm), O.oracle_all2all(seqs, prm, threads=8))


def test_model_rejects_unsupported_params():
    with pytest.raises(ValueError):
        U.model_all2all(U.edge_set()[:2], dict(mqd=65))
    with pytest.raises(ValueError):
        U.model_all2all(U.edge_set()[:2], dict(aw=65))


def test_model_regions_stream_equals_calc_regions():
    """The alignment instantiation (RegionCoords fed with the factor stream, including the data_pos
    quirk of the gap fill) against the oracle's calc_regions, itself pinned by example/output/ani.aln.tsv."""
    _, ex = U.load_example()
    cases = [(r, q, None) for r in range(0, 12, 2) for q in range(12) if r != q]
    cases += [(r, q, dict(mal=15, msl=9, reg=60)) for r in (1, 5) for q in (0, 4, 6)]
    cases += [(r, q, dict(mrd=20, mqd=60)) for r in (3,) for q in (0, 1, 2)]
    for r, q, prm in cases:
        res, regs = U.model_pair_regions(ex[r], ex[q], prm)
        ores, oregs = O.oracle_pair(ex[r], ex[q], prm, want_regions=True)
        assert res == ores and np.array_equal(regs, oregs), (r, q, prm)
    E = U.edge_set()
    for r in range(len(E)):
        for q in range(len(E)):
            if r != q:
                res, regs = U.model_pair_regions(E[r], E[q])
                ores, oregs = O.oracle_pair(E[r], E[q], None, want_regions=True)
                assert res == ores and np.array_equal(regs, oregs), (r, q)


@pytest.mark.parametrize("words", [0, 1])
def test_lane_serial_policy(words):
    """LaneWave (lzani_core.h): the same machine driven by one lane per pair with word-parallel bit tricks
    and a second (msl) index -- a second, independent host model (round 1 also ran it as a thread-per-pair GPU kernel)."""
    import ctypes as C
    lib = U.model_lib()

    def lane(seqs, prm=None):
        seqs, ptrs, lens = O._seq_table(seqs)
        n = len(seqs)
        out = np.zeros((n, n, 3), dtype=np.int32)
        assert lib.model_lane_all2all(n, ptrs, O._ptr(lens), O.params_array(prm), words, O._ptr(out)) == 0
        return out
    _, ex = U.load_example()
    assert np.array_equal(lane(ex[:8]), O.oracle_all2all(ex[:8], None, threads=8))
    E = U.edge_set()
    for prm in (None, dict(mal=15, msl=9, reg=60), dict(mrd=0), dict(ar=1), dict(aw=64, am=20), dict(mqd=64, mrd=64),
                dict(mqd=0), dict(mal=20, msl=12), dict(ar=0), dict(msl=1, mal=4)):
        assert np.array_equal(lane(E, prm), O.oracle_all2all(E, prm, threads=8)), prm


def test_differential_fuzz_models_vs_oracle():
    """Seeded random parameters and sequences: both host models of the kernel formulation against the
    oracle (and the oracle against the reference build where present).  This is the test that found the
    mrd = 0 case (no pad between the forward text and its reverse complement)."""
    import ctypes as C
    lib = U.model_lib()
    st = SG.Stream(2024)
    for it in range(250):
        prm, seqs = U.fuzz_case(st)
        want = O.oracle_all2all(seqs, prm, threads=4)
        assert np.array_equal(U.model_all2all(seqs, prm), want), (it, prm)
        s, ptrs, lens = O._seq_table(seqs)
        out = np.zeros((len(s), len(s), 3), dtype=np.int32)
        assert lib.model_lane_all2all(len(s), ptrs, O._ptr(lens), O.params_array(prm), it & 1, O._ptr(out)) == 0
        assert np.array_equal(out, want), (it, prm)
        if O.lib_ref() is not None and it % 4 == 0 and prm["msl"] <= 11:     # the reference's 4^msl table: msl >= 16 overflows it
            assert np.array_equal(O.ref_all2all(seqs, prm, threads=4), want), (it, prm)
        r, q = it % 5, (it // 5) % 5
        if r != q:                                                          # streaming calc_regions too
            res, regs = U.model_pair_regions(seqs[r], seqs[q], prm)
            ores, oregs = O.oracle_pair(seqs[r], seqs[q], prm, want_regions=True)
            assert res == ores and np.array_equal(regs, oregs), (it, prm, r, q)


def _queue_model(seqs, prm=None):
    """The anchor-queue formulation of find_event (tests/model/queue_wave.h); None where it does not apply
    (no k-mer words, or tags that do not identify the k-mer)."""
    lib = U.model_lib()
    s, ptrs, lens = O._seq_table(seqs)
    out = np.zeros((len(s), len(s), 3), dtype=np.int32)
    rc = lib.model_queue_all2all(len(s), ptrs, O._ptr(lens), O.params_array(prm), O._ptr(out))
    return out if rc == 0 else None


def test_anchor_queue_formulation():
    """Candidates detected ahead in chunks, resolved lane-serially with a 32-symbol cap, tracking rounds over the
    tracking steps only and jumps in lost mode: the same events as the reference's step-by-step scan."""
    _, ex = U.load_example()
    assert np.array_equal(_queue_model(ex), O.oracle_all2all(ex, None, threads=8))
    E = U.edge_set()
    done = 0
    for prm in list(U.VARIANTS.values()) + EXTRA:
        got = _queue_model(E, prm)
        if got is not None:
            assert np.array_equal(got, O.oracle_all2all(E, prm, threads=8)), prm
            done += 1
    assert done >= 10
    _, seqs = SG.make_set(12, 3, lmin=5000, lmax=9000, fam=4)
    seqs[2] = np.concatenate([seqs[2][:900], np.full(40, 5, np.uint8), seqs[2][900:]])
    for prm in (None, dict(mal=15, msl=9, reg=60), dict(mrd=0), dict(mrd=2, mqd=2), dict(mal=9, msl=12, mqd=30)):    # the last: mal < msl
        assert np.array_equal(_queue_model(seqs, prm), O.oracle_all2all(seqs, prm, threads=8)), prm
    st = SG.Stream(31337)
    done = 0
    for it in range(200):
        prm, seqs = U.fuzz_case(st)
        got = _queue_model(seqs, prm)
        if got is not None:
            assert np.array_equal(got, O.oracle_all2all(seqs, prm, threads=4)), (it, prm)
            done += 1
    assert done > 100


def test_ext_record_narrow_form_equals_wide_form():
    """refill's null-extension records (null_ext_record) read 16 symbols per side when aw <= 15 (32-bit arithmetic); the
    32-symbol form is the statement.  Same record for every candidate triple: random positions, positions at both ends
    of both strands and of the query, genomes with N runs (the N-mask branch), several (aw, am, ar)."""
    import ctypes as C
    lib = U.model_lib()
    st = SG.Stream(2024)
    for it in range(12):
        L1, L2 = st.randint(200, 3000), st.randint(200, 3000)
        r = (st.u64(L1) % np.uint64(4)).astype(np.uint8)
        q = SG.mutate(r, 0.05 + 0.2 * st.one(), st) if it % 2 else (st.u64(L2) % np.uint64(4)).astype(np.uint8)
        q = np.ascontiguousarray(q).copy()
        if it % 3 == 0:
            a = st.randint(0, len(q) - 40); q[a:a + st.randint(1, 30)] = 5
            r = r.copy(); a = st.randint(0, len(r) - 40); r[a:a + st.randint(1, 30)] = 5
        prm = dict(O.DEFAULTS) if hasattr(O, "DEFAULTS") else dict(mal=11, msl=7, mrd=40, mqd=40, reg=35, aw=15, am=7, ar=3)
        if it % 4 == 1: prm.update(aw=10, am=3, ar=5)
        if it % 4 == 2: prm.update(aw=15, am=2, ar=1)
        if it % 4 == 3: prm.update(aw=7, am=6, ar=3, mrd=25)
        T = 2 * len(r) + 3 * prm["mrd"]
        n = 4000
        qp = np.array([st.randint(0, len(q) + prm["mrd"] - 1) for _ in range(n)], dtype=np.int32)
        rp = np.array([st.randint(0, T - 1) for _ in range(n)], dtype=np.int32)
        al = np.array([st.randint(0, 40) for _ in range(n)], dtype=np.int32)
        edge = [0, 1, 15, 16, 17, 31, 32, 33, len(r) - 33, len(r) - 16, len(r) - 1, len(r), len(r) + 2 * prm["mrd"] - 1, len(r) + 2 * prm["mrd"],
                len(r) + 2 * prm["mrd"] + 16, len(r) + 2 * prm["mrd"] + 33, T - 40, T - 17, T - 1]
        for k, e in enumerate(edge):
            rp[k] = max(0, min(T - 1, e)); qp[k + 32] = max(0, min(len(q) + prm["mrd"] - 1, e if e < len(q) else len(q) - (k % 40)))
        bad = lib.model_ext_records_agree(O._ptr(r), len(r), O._ptr(q), len(q), O.params_array(prm), n, O._ptr(qp), O._ptr(rp), O._ptr(al))
        assert bad == 0, (it, prm, bad)


def test_split_pairs_model_against_oracle():
    """One pair by several segments (lzani_core.h: SplitStart / run_checkpoint / run_segment / split_stitch, the formulation
    behind lzani_kernels_split.h) through the host model: cuts every few hundred to few thousand query positions, every
    hand-over an equality of states, void segments run again with their cut disabled -- results equal to the oracle's for
    related and unrelated pairs, default and long-genome parameters, and every pair gets through the stitch."""
    for seed, (lmin, lmax, fam, dmax) in enumerate([(5000, 9000, 4, 0.15), (8000, 12000, 6, 0.05), (3000, 4000, 1, 0.15)]):
        _, seqs = SG.make_set(8, 60 + seed, lmin=lmin, lmax=lmax, fam=fam, dmax=dmax)
        for prm in (None, dict(mal=15, msl=9, reg=60), dict(mqd=20, mrd=30, reg=20)):
            want = O.oracle_all2all(seqs, prm, threads=8)
            for seglen in (300, 1000, 2500):
                got, st = U.model_split_all2all(seqs, prm, seglen)
                assert np.array_equal(got, want), (seed, prm, seglen)
                assert st[0] == len(seqs) * (len(seqs) - 1) and st[3] < 1000000, (seed, prm, seglen, st.tolist())     # every pair stitched, none scanned whole
    E = U.edge_set()
    for prm in (None, dict(mal=15, msl=9, reg=60), dict(mrd=0), dict(reg=1), dict(aw=64, am=20)):
        got, _ = U.model_split_all2all(E, prm, 200)
        assert np.array_equal(got, O.oracle_all2all(E, prm, threads=8)), prm


def test_split_pairs_model_fuzz():
    st = SG.Stream(4242)
    for it in range(400):
        prm, seqs = U.fuzz_case(st)
        want = O.oracle_all2all(seqs, prm, threads=4)
        got, _ = U.model_split_all2all(seqs, prm, 64 + 37 * (it % 9))
        assert np.array_equal(got, want), (it, prm)
