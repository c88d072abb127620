#!/usr/bin/env python3
"""Generate tests/golden/ref_vectors.json with the reference's own CParser (oracle/_ref).

TEST INFRASTRUCTURE ONLY.  Run in the build container (needs /root/reference to build
oracle/_ref):   make -C oracle ref && python oracle/make_goldens.py
Every vector is (inputs named by fixture/seed, expected int triples); no reference code is stored.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in ("oracle", "tools", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np  # noqa: E402
import oracle as O  # noqa: E402
import synth_genomes as SG  # noqa: E402
import util as U  # noqa: E402

EXTRA = {"mrd0": dict(mrd=0), "ar1": dict(ar=1), "aw64_am20": dict(aw=64, am=20),
         "mqd64_mrd64": dict(mqd=64, mrd=64), "mqd0": dict(mqd=0), "reg1": dict(reg=1), "am0": dict(am=0)}


def main():
    assert O.lib_ref() is not None, "build oracle/_ref first (make -C oracle ref)"
    out = {"generator": "oracle/make_goldens.py", "source": "CParser of /root/reference (LZ-ANI 1.2.3) via oracle/ref_driver.cpp",
           "layout": "res[r][q] = [sym_in_matches, sym_in_literals, no_components] of parse(query=q, ref=r); diagonal zero",
           "sets": {}}
    _, ex = U.load_example()
    _, vir = U.load_vir61()
    edge = U.edge_set()
    _, syn = SG.make_set(24, 11, lmin=6000, lmax=9000, fam=6)
    for name, prm in U.VARIANTS.items():
        out["sets"][f"example/{name}"] = dict(params=prm, res=O.ref_all2all(ex, prm, threads=8).tolist())
        out["sets"][f"edge/{name}"] = dict(params=prm, res=O.ref_all2all(edge, prm, threads=8).tolist())
        out["sets"][f"synth24/{name}"] = dict(params=prm, res=O.ref_all2all(syn, prm, threads=8).tolist())
    for name, prm in EXTRA.items():
        out["sets"][f"edge/{name}"] = dict(params=prm, res=O.ref_all2all(edge, prm, threads=8).tolist())
        out["sets"][f"synth24/{name}"] = dict(params=prm, res=O.ref_all2all(syn, prm, threads=8).tolist())
    out["sets"]["vir61/default"] = dict(params={}, res=O.ref_all2all(vir, None, threads=8).tolist())
    # per-region vectors (CParser::get_parsing) for the 12 example genomes, default parameters
    regs = {}
    for r in range(len(ex)):
        for q in range(len(ex)):
            if r != q:
                _, rg = O.ref_pair(ex[r], ex[q], None, want_regions=True)
                if len(rg):
                    regs[f"{r},{q}"] = rg.tolist()
    out["regions_example_default"] = regs
    path = os.path.join(ROOT, "tests", "golden", "ref_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;", len(out["sets"]), "sets")


if __name__ == "__main__":
    main()
