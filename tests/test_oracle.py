"""CPU: pin the oracle (oracle/lzani_oracle.c) before it is trusted as the parity checker.

1. against the reference's own golden files (test/vir61.ani.tsv, example/output/*.tsv), rebuilt
   byte for byte from the oracle's integers through the restated emit rule;
2. against vectors produced by the reference's CParser itself (tests/golden/ref_vectors.json,
   generator oracle/make_goldens.py);
3. live against oracle/_ref where it has been built (this container; it also travels to the GPU box).
"""
import json
import os

import numpy as np
import pytest

import oracle as O
import synth_genomes as SG
import util as U


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(U.GOLD, "ref_vectors.json")) as f:
        return json.load(f)


def _inputs(setname):
    if setname == "example":
        return U.load_example()[1]
    if setname == "edge":
        return U.edge_set()
    if setname == "vir61":
        return U.load_vir61()[1]
    if setname == "synth24":
        return SG.make_set(24, 11, lmin=6000, lmax=9000, fam=6)[1]
    raise KeyError(setname)


def test_vir61_golden_tsv_byte_exact():
    names, seqs = U.reorder(*U.load_vir61())
    res = O.oracle_all2all(seqs, None, threads=8)
    txt = U.emit_tsv(names, [len(s) for s in seqs], res, U.STANDARD)
    assert txt == open(os.path.join(U.GOLD, "vir61.ani.tsv")).read()
    ids = "id\tseq_len\tno_parts\n" + "".join(f"{n}\t{len(s)}\t1\n" for n, s in zip(names, seqs))
    assert ids == open(os.path.join(U.GOLD, "vir61.ani.ids.tsv")).read()


def test_example_golden_tsv_byte_exact():
    names, seqs = U.reorder(*U.load_example())
    res = O.oracle_all2all(seqs, None, threads=8)
    txt = U.emit_tsv(names, [len(s) for s in seqs], res, U.STANDARD)
    assert txt == open(os.path.join(U.GOLD, "example", "ani.tsv")).read()


def test_example_alignment_golden_multiset():
    """example/output/ani.aln.tsv: per-region rows of calc_regions + store_alignment (row order is
    thread-schedule dependent in the reference, so compare as a multiset)."""
    names, seqs = U.reorder(*U.load_example())
    mrd = 40
    rows = []
    for r in range(len(seqs)):
        for q in range(len(seqs)):
            if r == q:
                continue
            _, regs = O.oracle_pair(seqs[r], seqs[q], None, want_regions=True)
            rc_corr = 2 * len(seqs[r]) + 2 * mrd + 1
            for g in regs:
                rs, re, ss, se, nm, nmm = [int(x) for x in g]
                ln = se - ss
                if rs < len(seqs[r]):
                    a, b = 1 + rs, re
                else:
                    a, b = rc_corr - (1 + rs), rc_corr - re
                rows.append("\t".join([names[q], names[r], U.real_to_str(100.0 * nm / ln, 6), str(ln), str(1 + ss),
                                       str(se), str(a), str(b), str(nm), str(nmm)]))
    gold = open(os.path.join(U.GOLD, "example", "ani.aln.tsv")).read().split("\n")
    assert gold[0].startswith("query\treference\tpident")
    assert sorted(rows) == sorted(x for x in gold[1:] if x)


def test_reference_vectors(vectors):
    for key, item in vectors["sets"].items():
        setname = key.split("/")[0]
        got = O.oracle_all2all(_inputs(setname), item["params"], threads=8)
        want = np.array(item["res"], dtype=np.int32)
        bad = np.argwhere((got != want).any(axis=2))
        assert len(bad) == 0, f"{key}: {len(bad)} pairs differ, first {bad[:3].tolist()}"


def test_reference_region_vectors(vectors):
    ex = U.load_example()[1]
    for key, want in vectors["regions_example_default"].items():
        r, q = (int(x) for x in key.split(","))
        _, regs = O.oracle_pair(ex[r], ex[q], None, want_regions=True)
        assert regs.tolist() == want, key


def test_edge_values_known():
    """A few hand-checkable answers (identical copy, reverse complement, too-short inputs)."""
    e = U.edge_set()
    assert O.oracle_pair(e[0], e[1]) == (3000, 0, 1)          # identical copy
    assert O.oracle_pair(e[0], e[2]) == (3000, 0, 1)          # reverse complement found on the RC half
    assert O.oracle_pair(e[0], e[6]) == (0, 0, 0)             # all-N query
    assert O.oracle_pair(e[0], e[9]) == (0, 0, 0)             # empty query
    assert O.oracle_pair(e[9], e[0]) == (0, 0, 0)             # empty reference


@pytest.mark.skipif(O.lib_ref() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_live_against_reference_build():
    _, seqs = SG.make_set(30, 21, lmin=4000, lmax=12000, fam=5)
    seqs[3] = np.concatenate([seqs[3][:2000], np.full(30, 5, np.uint8), seqs[3][2000:]])
    for prm in (None, dict(mal=15, msl=9, reg=60), dict(mrd=25, mqd=55, aw=20, am=9, ar=2)):
        a = O.oracle_all2all(seqs, prm, threads=8)
        b = O.ref_all2all(seqs, prm, threads=8)
        assert np.array_equal(a, b), prm
