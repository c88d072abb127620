#!/bin/bash
# GPU box, round 4 run 13: the 1,024-symbol extension step: parity (long genomes, fuzz), A/B against the build without it
set -o pipefail
mkdir -p gpurun_out
REL="--workload related --genomes 20000 --fam 50 --seed 1"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "reference_vectors or vir61 or full_size_properties_1000 or filtered_heavy or fuzz or mixed_n or sparse_rows or bacterial or config4 or null_chain_long or natural_trigger" > gpurun_out/r4_run13_pytest.log 2>&1 || { tail -40 gpurun_out/r4_run13_pytest.log; exit 1; }
tail -2 gpurun_out/r4_run13_pytest.log
timeout -k 10 300 python tools/fuzz_gpu.py 777 150 medium > gpurun_out/r4_fuzz_medium13.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_medium13.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_medium13.log
timeout -k 10 300 python tools/fuzz_gpu.py 778 100 large > gpurun_out/r4_fuzz_large13.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_large13.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_large13.log
for LIB in "" build/exp/nowide.so; do
for D in 0.15 0.05; do
LZANI_LIB=${LIB:+$PWD/$LIB} timeout -k 10 600 python bench.py $REL --dmax $D --steps 4 --warmup 1 --cpu-sample 0 > gpurun_out/r4_related_f_$D.json 2> gpurun_out/r4_related_f_$D.err || { tail -5 gpurun_out/r4_related_f_$D.err; exit 1; }
python - $D "${LIB:-shipped}" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4_related_f_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%s related <= %s: %.3f M pairs/s, kernel %.1f ms (%.3f M pairs/s), parity %s" % (sys.argv[2], sys.argv[1], d["value"]/1e6, r["avg_launch_ms"], d["config"]["pairs_per_step"]/r["avg_launch_ms"]/1e3, d.get("parity_on_last_slab")))
PY
done
done
bash tools/c4_bench.sh 128
bash tools/c4_bench.sh 128 LZANI_LIB=$PWD/build/exp/nowide.so
