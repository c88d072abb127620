#!/bin/bash
# GPU box, round 4 run 4: candidate bitmaps for filtered rows (LZANI_PM_MIN_SHARE) with the stretch path reading them
set -o pipefail
mkdir -p gpurun_out
REL="--workload related --genomes 20000 --fam 50 --dmax 0.15 --seed 1"
for SH in 48 8; do
LZANI_PM_MIN_SHARE=$SH timeout -k 10 600 python bench.py $REL --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/r4_related_share$SH.json 2> gpurun_out/r4_related_share$SH.err || { tail -5 gpurun_out/r4_related_share$SH.err; exit 1; }
python - $SH <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4_related_share%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("related, min share %s: %.3f M pairs/s, kernel %.1f ms, cand %.1f ms, index %.1f ms per %d pairs, bitmaps %d, parity %s" % (sys.argv[1], d["value"]/1e6, r["avg_launch_ms"], r["candidate_stage_ms_per_step"], r["index_build_ms_per_step"], d["config"]["pairs_per_step"], d["config"]["index_form"]["candidate_bitmaps_from_presence_matrix"], d.get("parity_on_last_slab")))
PY
done
