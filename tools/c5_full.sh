#!/bin/bash
# GPU box: BASELINE configs[4] on one GPU -- N synthetic ~40 kbp genomes in heavy-tailed families, a synthetic kmer-db
# file, `lz-ani all2all --flt-kmerdb f 0.3` end to end, then sampled TSV rows against the oracle.
# Usage: tools/c5_full.sh [n_genomes=100000] [seed=4]
set -o pipefail
N=${1:-100000}; SEED=${2:-4}
ROOT=$(pwd)
D=${TMPDIR:-/tmp}/c5_$$
mkdir -p "$D"
g++ -O2 -std=c++17 -o "$D/synth5" "$ROOT/tools/synth5.cpp" || exit 1
/usr/bin/env time -v true 2>/dev/null
T0=$(date +%s.%N)
"$D/synth5" "$N" "$SEED" 36000 44000 "$D/in.fna" "$D/in.flt" "$D/in.bin" 1000 0.3 > "$D/gen.json" || exit 1
T1=$(date +%s.%N)
cat "$D/gen.json"; echo "generator: $(echo "$T1 - $T0" | bc -l 2>/dev/null || python3 -c "print($T1-$T0)") s"
ls -la "$D"/in.* | awk '{print $5, $9}'
"$ROOT/lz-ani_amd/host/lz-ani" all2all --in-fasta "$D/in.fna" --out "$D/out.tsv" --flt-kmerdb "$D/in.flt" 0.3 -V 2 \
    --out-format query,reference,nt_match,nt_mismatch,num_alns > "$D/out.log" 2> "$D/err.log"
echo "exit $?"
grep -v "^\s*[0-9]*%" "$D/err.log" | tail -16
wc -lc "$D/out.tsv" | awk '{print "tsv lines", $1, "bytes", $2}'
python3 - "$D" "$N" <<'PY'
import json, os, sys
import numpy as np
root = os.getcwd()
for p in ("oracle", "tools"):
    sys.path.insert(0, os.path.join(root, p))
import oracle as O
d, n = sys.argv[1], int(sys.argv[2])
info = json.load(open(os.path.join(d, "gen.json")))
hdr = np.fromfile(os.path.join(d, "in.bin"), dtype=np.uint64, count=n + 2)
off = hdr[1:].astype(np.int64)
codes = np.memmap(os.path.join(d, "in.bin"), dtype=np.uint8, mode="r", offset=8 * (n + 2))
seq = lambda i: np.array(codes[off[i]:off[i + 1]])
lines = 0
picks = []
with open(os.path.join(d, "out.tsv")) as f:
    f.readline()
    for k, ln in enumerate(f):
        lines += 1
        if (k * 2654435761) % 2**32 < 2**32 // 100000 and len(picks) < 600:
            qn, rn, mat, lit, aln = ln.rstrip("\n").split("\t")
            picks.append((int(rn[1:7]), int(qn[1:7]), int(mat), int(lit), int(aln)))
ids = sorted({x for p in picks for x in p[:2]})
local = {g: k for k, g in enumerate(ids)}
sub = [seq(g) for g in ids]
rr = np.array([local[p[0]] for p in picks], np.uint32)
qq = np.array([local[p[1]] for p in picks], np.uint32)
if O.lib_ref() is not None:
    want = O.ref_rows(sub, rr, np.arange(len(picks) + 1, dtype=np.uint64), qq, None, threads=len(os.sched_getaffinity(0)))
else:
    want = np.array([O.oracle_pair(sub[r], sub[q]) for r, q in zip(rr, qq)], dtype=np.int32)
bad = sum(tuple(int(x) for x in want[k]) != p[2:] for k, p in enumerate(picks))
checked = len(picks)
print("tsv data lines", lines, "expected", 2 * info["pairs_kept"], "sampled rows checked", checked, "differing", bad)
PY
rm -rf "$D"
