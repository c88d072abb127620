#!/bin/bash
# GPU box, round 4 run 5: the stretch chain -- parity first (tests + fuzz), then the related rates and the path counters
set -o pipefail
mkdir -p gpurun_out
REL="--workload related --genomes 20000 --fam 50 --seed 1"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "reference_vectors or vir61 or full_size_properties_1000 or filtered_heavy or fuzz or mixed_n or sparse_rows" > gpurun_out/r4_run5_pytest.log 2>&1 || { tail -40 gpurun_out/r4_run5_pytest.log; exit 1; }
tail -2 gpurun_out/r4_run5_pytest.log
timeout -k 10 300 python tools/fuzz_gpu.py 99 150 medium > gpurun_out/r4_fuzz_medium5.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_medium5.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_medium5.log
for D in 0.15 0.05; do
timeout -k 10 600 python bench.py $REL --dmax $D --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/r4_related_sc_$D.json 2> gpurun_out/r4_related_sc_$D.err || { tail -5 gpurun_out/r4_related_sc_$D.err; exit 1; }
python - $D <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4_related_sc_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("related <= %s: %.3f M pairs/s, kernel %.1f ms per %d pairs (%.3f M pairs/s), parity %s" % (sys.argv[1], d["value"]/1e6, r["avg_launch_ms"], d["config"]["pairs_per_step"], d["config"]["pairs_per_step"]/r["avg_launch_ms"]/1e3, d.get("parity_on_last_slab")))
PY
done
LZANI_LIB=$PWD/build/exp/paths.so timeout -k 10 300 python bench.py --workload related --genomes 8000 --fam 50 --dmax 0.15 --seed 1 --steps 1 --warmup 1 --cpu-sample 0 --no-check > gpurun_out/r4_paths5.json 2> gpurun_out/r4_paths5.log || { tail -5 gpurun_out/r4_paths5.log; exit 1; }
grep "lzani paths" gpurun_out/r4_paths5.log | tail -1 | tr ';' '\n' | grep -E "stretch|events|chunks|light"
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --cpu-sample 0 > gpurun_out/r4_sc_base_line.json 2> gpurun_out/r4_sc_base_line.err || { tail -5 gpurun_out/r4_sc_base_line.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_sc_base_line.json").read().strip().splitlines()[-1])
print("base: %.3f M pairs/s, kernel %.1f ms, parity %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d.get("parity_on_last_slab")))
PY
