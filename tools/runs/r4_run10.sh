#!/bin/bash
# GPU box, round 4 run 10: the null chain's seed event at the queued candidate's own step: parity (incl. fuzz), rates
set -o pipefail
mkdir -p gpurun_out
REL="--workload related --genomes 20000 --fam 50 --seed 1"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "reference_vectors or vir61 or full_size_properties_1000 or full_size_10k or filtered_heavy or fuzz or mixed_n or sparse_rows or bacterial or null_chain_long or presence_matrix_candidates" > gpurun_out/r4_run10_pytest.log 2>&1 || { tail -40 gpurun_out/r4_run10_pytest.log; exit 1; }
tail -2 gpurun_out/r4_run10_pytest.log
timeout -k 10 300 python tools/fuzz_gpu.py 555 150 medium > gpurun_out/r4_fuzz_medium10.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_medium10.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_medium10.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r4_e_base_line.json 2> gpurun_out/r4_e_base_line.err || { tail -5 gpurun_out/r4_e_base_line.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_e_base_line.json").read().strip().splitlines()[-1])
print("base: %.3f M pairs/s, kernel %.1f ms, parity %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d.get("parity_on_last_slab")))
PY
for D in 0.15 0.05; do
timeout -k 10 600 python bench.py $REL --dmax $D --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/r4_related_e_$D.json 2> gpurun_out/r4_related_e_$D.err || { tail -5 gpurun_out/r4_related_e_$D.err; exit 1; }
python - $D <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4_related_e_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("related <= %s: %.3f M pairs/s, kernel %.1f ms per %d pairs (%.3f M pairs/s), parity %s" % (sys.argv[1], d["value"]/1e6, r["avg_launch_ms"], d["config"]["pairs_per_step"], d["config"]["pairs_per_step"]/r["avg_launch_ms"]/1e3, d.get("parity_on_last_slab")))
PY
done
