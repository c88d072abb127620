#!/bin/bash
# GPU box, round 4 run 14: candidate bitmaps (presence matrix) for the filtered rows of the related workload:
# probe form (shipped heuristic) vs bitmaps without / with the stretch chain (LZANI_STRETCH_DENSE build)
set -o pipefail
mkdir -p gpurun_out
REL="--workload related --genomes 20000 --fam 50 --seed 1"
run() {   # name lib share dmax
    LZANI_LIB=${2:+$PWD/$2} LZANI_PM_MIN_SHARE=$3 timeout -k 10 600 python bench.py $REL --dmax $4 --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/r14_$1_$4.json 2> gpurun_out/r14_$1_$4.err || { tail -5 gpurun_out/r14_$1_$4.err; exit 1; }
    python - $1 $4 <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r14_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
r = d["roofline"]
print("%-14s d<=%s: %.3f M pairs/s, kernel %.1f ms, cand %.1f ms, index %.1f ms per %d pairs, bitmaps %d, parity %s" % (sys.argv[1], sys.argv[2], d["value"]/1e6, r["avg_launch_ms"], r["candidate_stage_ms_per_step"], r["index_build_ms_per_step"], d["config"]["pairs_per_step"], d["config"]["index_form"]["candidate_bitmaps_from_presence_matrix"], d.get("parity_on_last_slab")))
PY
}
for D in 0.15 0.05; do
run probe "" 48 $D || exit 1
run pm_lean "" 4 $D || exit 1
run pm_stretch build/exp/sdense.so 4 $D || exit 1
done
