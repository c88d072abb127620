#!/bin/bash
# GPU box, round 4 run 33: k_pm_cand with 1,024 threads a block as the default: the candidate-stage tests, the bench, 128 x 5 Mbp
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "presence or bitmap or lists or from_index or fuzz or vir61 or reference_vectors or natural_trigger or mid_size or split_over or run_time" > gpurun_out/r4_run33_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run33_pytest.log; exit 1; }
tail -1 gpurun_out/r4_run33_pytest.log
timeout -k 10 400 python bench.py > gpurun_out/r4_final_bench_default_line.json 2> gpurun_out/r4_final_bench.err || { tail -5 gpurun_out/r4_final_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_final_bench_default_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("default bench: %.3f M pairs/s, %.1f ms/step, kernel %.1f ms, cand %.1f ms, frac %.4f, cpu %.0f pairs/s, parity %s / %s" % (d["value"]/1e6, d["ms_per_step"], r["avg_launch_ms"], r["candidate_stage_ms_per_step"], r["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["parity_on_sample"], d["parity_on_last_slab"]))
PY
bash tools/c4_bench.sh 128
