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
