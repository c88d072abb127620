#!/bin/bash
# GPU box, round 4 run 3: run-time compiled kernels + the related-stretch path: parity first, then rates
set -o pipefail
mkdir -p gpurun_out
REL="--workload related --genomes 20000 --fam 50 --dmax 0.15 --seed 1"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "run_time_compiled or reference_vectors or vir61 or full_size_properties_1000 or filtered_heavy or fuzz" > gpurun_out/r4_run3_pytest.log 2>&1 || { tail -40 gpurun_out/r4_run3_pytest.log; exit 1; }
tail -2 gpurun_out/r4_run3_pytest.log
timeout -k 10 600 python bench.py $REL --steps 5 --warmup 1 > gpurun_out/r4_related_stretch_line.json 2> gpurun_out/r4_related_stretch_line.err || { tail -5 gpurun_out/r4_related_stretch_line.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_related_stretch_line.json").read().strip().splitlines()[-1])
print("related: %.3f M pairs/s, kernel %.1f ms per %d pairs, frac %.4f, parity %s / %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d["config"]["pairs_per_step"], d["roofline"]["frac"], d.get("parity_on_last_slab"), d["cpu_baseline"]["parity_on_sample"]))
PY
timeout -k 10 600 python bench.py --workload related --genomes 20000 --fam 50 --dmax 0.05 --seed 1 --steps 5 --warmup 1 --cpu-sample 0 > gpurun_out/r4_related5_stretch_line.json 2> gpurun_out/r4_related5_stretch_line.err || { tail -5 gpurun_out/r4_related5_stretch_line.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_related5_stretch_line.json").read().strip().splitlines()[-1])
print("related <= 5%%: %.3f M pairs/s, kernel %.1f ms per %d pairs, parity %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d["config"]["pairs_per_step"], d.get("parity_on_last_slab")))
PY
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/r4_stretch_base_line.json 2> gpurun_out/r4_stretch_base_line.err || { tail -5 gpurun_out/r4_stretch_base_line.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_stretch_base_line.json").read().strip().splitlines()[-1])
print("base: %.3f M pairs/s, kernel %.1f ms, parity %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d.get("parity_on_last_slab")))
PY
for P in reg=36 am=6; do
timeout -k 10 300 python bench.py --params $P --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r4_rtc_${P//[=,]/_}_line.json 2> gpurun_out/r4_rtc_${P//[=,]/_}.err || { tail -5 gpurun_out/r4_rtc_${P//[=,]/_}.err; exit 1; }
python - "$P" gpurun_out/r4_rtc_${P//[=,]/_}_line.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%s (run-time compiled): %.3f M pairs/s, kernel %.1f ms, parity %s" % (sys.argv[1], d["value"]/1e6, d["roofline"]["avg_launch_ms"], d.get("parity_on_last_slab")))
PY
done
timeout -k 10 400 python tools/fuzz_gpu.py 4242 240 rtc 20 > gpurun_out/r4_rtc_fuzz.log 2>&1 || { tail -20 gpurun_out/r4_rtc_fuzz.log; exit 1; }
tail -2 gpurun_out/r4_rtc_fuzz.log
timeout -k 10 300 python tools/fuzz_gpu.py 77 200 medium > gpurun_out/r4_fuzz_medium.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_medium.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_medium.log
