#!/bin/bash
# GPU box: BASELINE configs[3] at full size on one GPU -- N synthetic ~5 Mbp genomes in families of 10 (d ~ U(0.005, 0.08)),
# `lz-ani all2all --mal 15 --msl 9 --reg 60` end to end, then sampled TSV rows against the oracle.
# Usage: tools/c4_full.sh [n_genomes=1000] [seed=3] [rows_checked=200]
set -o pipefail
N=${1:-1000}; SEED=${2:-3}; CHECK=${3:-200}
ROOT=$(pwd)
D=${TMPDIR:-/tmp}/c4_$$
mkdir -p "$D"
g++ -O2 -std=c++17 -o "$D/synth5" "$ROOT/tools/synth5.cpp" || exit 1
T0=$(date +%s.%N)
"$D/synth5" "$N" "$SEED" 4500000 5500000 "$D/in.fna" "$D/in.flt" "$D/in.bin" 1000 0.3 10 0.005 0.08 > "$D/gen.json" || exit 1
T1=$(date +%s.%N)
cat "$D/gen.json"; python3 -c "print('generator: %.1f s' % ($T1 - $T0))"
ls -la "$D"/in.fna | awk '{print "fasta bytes", $5}'
rm -f "$D/in.flt"
"$ROOT/lz-ani_amd/host/lz-ani" all2all --in-fasta "$D/in.fna" --out "$D/out.tsv" --mal 15 --msl 9 --reg 60 -V 2 \
    --out-format query,reference,nt_match,nt_mismatch,num_alns > "$D/out.log" 2> "$D/err.log"
echo "exit $?"
grep -v "^\s*[0-9]*%" "$D/err.log" | tail -14
wc -lc "$D/out.tsv" | awk '{print "tsv lines", $1, "bytes", $2}'
rm -f "$D/in.fna"
python3 - "$D" "$N" "$CHECK" <<'PY'
import json, os, sys
import numpy as np
root = os.getcwd()
for p in ("oracle", "tools"):
    sys.path.insert(0, os.path.join(root, p))
import oracle as O
d, n, check = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
hdr = np.fromfile(os.path.join(d, "in.bin"), dtype=np.uint64, count=n + 2)
off = hdr[1:].astype(np.int64)
codes = np.memmap(os.path.join(d, "in.bin"), dtype=np.uint8, mode="r", offset=8 * (n + 2))
seq = lambda i: np.array(codes[off[i]:off[i + 1]])
prm = dict(mal=15, msl=9, reg=60)
lines = 0
want_rel = check // 3
picks = []                                   # (ref, query, mat, lit, aln) of the sampled rows: a third of them related pairs
with open(os.path.join(d, "out.tsv")) as f:
    f.readline()
    for k, ln in enumerate(f):
        lines += 1
        if len(picks) >= check:
            continue
        qn, rn, mat, lit, aln = ln.rstrip("\n").split("\t")
        qi, ri = int(qn[1:7]), int(rn[1:7])
        related = qi // 10 == ri // 10
        if n * (n - 1) <= check or (related and want_rel > 0 and (k * 2654435761) % 2**32 < 2**32 // 40) or (not related and (k * 2654435761) % 2**32 < 2**32 // 3000):
            picks.append((ri, qi, int(mat), int(lit), int(aln)))
            want_rel -= related
# the sampled pairs through the reference's parser, all host threads at once (serially through the C restatement where
# oracle/_ref is not built)
ids = sorted({x for p in picks for x in p[:2]})
local = {g: k for k, g in enumerate(ids)}
sub = [seq(g) for g in ids]
rr = np.array([local[p[0]] for p in picks], np.uint32)
qq = np.array([local[p[1]] for p in picks], np.uint32)
if O.lib_ref() is not None:
    want = O.ref_rows(sub, rr, np.arange(len(picks) + 1, dtype=np.uint64), qq, prm, threads=len(os.sched_getaffinity(0)))
else:
    want = np.array([O.oracle_pair(sub[r], sub[q], prm) for r, q in zip(rr, qq)], dtype=np.int32)
bad = sum(tuple(int(x) for x in want[k]) != p[2:] for k, p in enumerate(picks))
checked = len(picks)
print("tsv data lines", lines, "expected", n * (n - 1), "sampled rows checked", checked, "differing", bad)
PY
rm -rf "$D"
