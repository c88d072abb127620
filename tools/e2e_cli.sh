#!/bin/bash
# GPU box: end-to-end wall time of the host binary on a synthetic multi-FASTA.  Usage: tools/e2e_cli.sh [n_genomes]
set -o pipefail
N=${1:-3000}
ROOT=$(pwd)
D=${TMPDIR:-/tmp}/e2e_$$
mkdir -p "$D"
python3 - "$N" "$D/in.fna" <<'PY'
import sys
sys.path.insert(0, "tools")
import numpy as np, synth_genomes as SG
n, path = int(sys.argv[1]), sys.argv[2]
names, seqs = SG.make_set(n, 1)
lut = np.frombuffer(b"ACGTNN", dtype=np.uint8)
with open(path, "wb") as f:
    for k, s in enumerate(seqs):
        f.write(b">g%06d synthetic\n" % k)
        a = lut[s]
        for o in range(0, len(a), 80):
            f.write(a[o:o + 80].tobytes() + b"\n")
print("fasta written:", n, "genomes")
PY
ls -la "$D/in.fna" | awk '{print "fasta bytes", $5}'
"$ROOT/lz-ani_amd/host/lz-ani" all2all --in-fasta "$D/in.fna" --out "$D/out.tsv" --out-ids "$D/ids.tsv" -V 2 > "$D/out.log" 2> "$D/err.log"
echo "exit $?"
tail -12 "$D/out.log"; tail -12 "$D/err.log"
wc -l "$D/out.tsv" | awk '{print "tsv lines", $1}'; head -3 "$D/out.tsv"
rm -rf "$D"
