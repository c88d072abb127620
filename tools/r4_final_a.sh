#!/bin/bash
# GPU box, end of round 4, part A: the whole -m gpu suite, the default bench line (driver contract), rocprofv3 kernel stats + PMC passes of it
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -q -m gpu > gpurun_out/r4_final_pytest.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4_final_pytest.log | tail -20; exit 1; }
tail -1 gpurun_out/r4_final_pytest.log
timeout -k 10 400 python bench.py > gpurun_out/r4_final_bench_default_line.json 2> gpurun_out/r4_final_bench.err || { tail -5 gpurun_out/r4_final_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_final_bench_default_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("default bench: %.3f M pairs/s, %.1f ms/step, kernel %.1f ms, cand %.1f ms, index %.2f ms, frac %.4f, cpu %.0f pairs/s on %d, parity %s / %s" % (d["value"]/1e6, d["ms_per_step"], r["avg_launch_ms"], r["candidate_stage_ms_per_step"], r["index_build_ms_per_step"], r["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["parity_on_sample"], d["parity_on_last_slab"]))
PY
bash tools/profile.sh r4_final || exit 1
ls gpurun_out/prof_r4_final/
