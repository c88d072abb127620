#!/bin/bash
# GPU box: end-to-end wall time of the host binary on a synthetic multi-FASTA.  Usage: tools/e2e_cli.sh [n_genomes] [seed] [extra lz-ani args]
set -o pipefail
N=${1:-3000}; SEED=${2:-2}; shift 2 2>/dev/null
ROOT=$(pwd)
D=${TMPDIR:-/tmp}/e2e_$$
mkdir -p "$D"
python3 - "$N" "$SEED" "$D/in.fna" <<'PY'
import sys
sys.path.insert(0, "tools")
import synth_genomes as SG
n, seed, path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
names, seqs = SG.make_set_cached(n, seed)
SG.write_fasta(path, names, seqs)
print("fasta written:", n, "genomes")
PY
ls -la "$D/in.fna" | awk '{print "fasta bytes", $5}'
df -h "$D" | tail -1
"$ROOT/lz-ani_amd/host/lz-ani" all2all --in-fasta "$D/in.fna" --out "$D/out.tsv" --out-ids "$D/ids.tsv" -V 2 "$@" > "$D/out.log" 2> "$D/err.log"
echo "exit $?"
tail -12 "$D/out.log"; tail -14 "$D/err.log"
wc -lc "$D/out.tsv" | awk '{print "tsv lines", $1, "bytes", $2}'; head -3 "$D/out.tsv"
rm -rf "$D"
