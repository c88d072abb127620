#!/bin/bash
# GPU box, round 4 run 8: the whole -m gpu suite, then the default bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_run8_pytest.log 2>&1 || { tail -40 gpurun_out/r4_run8_pytest.log; exit 1; }
tail -3 gpurun_out/r4_run8_pytest.log
timeout -k 10 300 python bench.py > gpurun_out/r4_d_base_line.json 2> gpurun_out/r4_d_base_line.err || { tail -5 gpurun_out/r4_d_base_line.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_d_base_line.json").read().strip().splitlines()[-1])
print("default bench: %.3f M pairs/s, kernel %.1f ms, cand %.1f, frac %.3f / %.3f, pcie %.2f M, parity %s / %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d["roofline"]["candidate_stage_ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_incl_candidate_stage"], d["pcie_inclusive_value"]/1e6, d.get("parity_on_last_slab"), d["cpu_baseline"]["parity_on_sample"]))
PY
