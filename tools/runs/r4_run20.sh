#!/bin/bash
# GPU box, round 4 run 20: rows that share their queries onto one XCD queue (filtered rows): parity of the filtered-row tests, then the related bench A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "filtered or sparse or lists or host_cli or mid_size or dist or group" > gpurun_out/r4_run20_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run20_pytest.log; exit 1; }
tail -2 gpurun_out/r4_run20_pytest.log
REL="--workload related --genomes 20000 --fam 50 --seed 1"
for D in 0.15 0.05; do
for V in 0 1; do
LZANI_FAMILY_QUEUES=$V timeout -k 10 600 python bench.py $REL --dmax $D --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/r20_rel_$D.json 2> gpurun_out/r20_rel_$D.err || { tail -5 gpurun_out/r20_rel_$D.err; exit 1; }
python - $V $D <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r20_rel_%s.json" % sys.argv[2]).read().strip().splitlines()[-1])
r = d["roofline"]
print("family queues=%s related d<=%s: %.3f M pairs/s, kernel %.1f ms (%.3f M pairs/s), index %.1f ms, host+rest %.1f ms per %d pairs, parity %s" % (sys.argv[1], sys.argv[2], d["value"]/1e6, r["avg_launch_ms"], d["config"]["pairs_per_step"]/r["avg_launch_ms"]/1e3, r["index_build_ms_per_step"], d["ms_per_step"] - r["avg_launch_ms"] - r["index_build_ms_per_step"], d["config"]["pairs_per_step"], d.get("parity_on_last_slab")))
PY
done
done
