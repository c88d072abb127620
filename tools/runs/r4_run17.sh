#!/bin/bash
# GPU box, round 4 run 17: k_idx_build2 (byte counters, two sweeps): the whole GPU suite, then the related bench A/B and 128 x 5 Mbp with the own sort
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -s -m gpu > gpurun_out/r4_run17_pytest.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4_run17_pytest.log | tail -20; exit 1; }
grep -E "passed|failed|mal 15|kbp" gpurun_out/r4_run17_pytest.log | tail -8
REL="--workload related --genomes 20000 --fam 50 --seed 1 --dmax 0.15"
for V in 0 1; do
LZANI_IDX_BUILD2=$V timeout -k 10 600 python bench.py $REL --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/r17_rel_$V.json 2> gpurun_out/r17_rel_$V.err || { tail -5 gpurun_out/r17_rel_$V.err; exit 1; }
python - $V <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r17_rel_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("build2=%s related d15: %.3f M pairs/s, kernel %.1f ms, index %.1f ms per %d pairs, parity %s" % (sys.argv[1], d["value"]/1e6, r["avg_launch_ms"], r["index_build_ms_per_step"], d["config"]["pairs_per_step"], d.get("parity_on_last_slab")))
PY
done
bash tools/c4_bench.sh 128 || exit 1
