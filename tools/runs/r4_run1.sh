#!/bin/bash
# GPU box, round 4 run 1: sanity of the new host code, baseline lines, related-pair evidence (kernel stats + PMC + stamps)
set -o pipefail
mkdir -p gpurun_out
REL="--workload related --genomes 20000 --fam 50 --dmax 0.15 --seed 1"
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "after_sparse_run or presence_matrix_candidates or sparse_rows_ragged" > gpurun_out/r4_run1_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run1_pytest.log; exit 1; }
tail -2 gpurun_out/r4_run1_pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/r4_base_line.json 2> gpurun_out/r4_base_line.err || { tail -5 gpurun_out/r4_base_line.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_base_line.json").read().strip().splitlines()[-1])
print("base: %.3f M pairs/s, kernel %.1f ms, cand %.1f ms, frac %.3f (incl cand %.3f), pcie %.3f M, parity %s / %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d["roofline"]["candidate_stage_ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_incl_candidate_stage"], d["pcie_inclusive_value"]/1e6, d.get("parity_on_last_slab"), d["cpu_baseline"]["parity_on_sample"]))
PY
timeout -k 10 300 python bench.py --params reg=36 --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r4_base_reg36_line.json 2> gpurun_out/r4_base_reg36.err || { tail -5 gpurun_out/r4_base_reg36.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_base_reg36_line.json").read().strip().splitlines()[-1])
print("reg=36 (DEFP=0): %.3f M pairs/s, kernel %.1f ms, parity %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d.get("parity_on_last_slab")))
PY
timeout -k 10 600 python bench.py $REL --steps 5 --warmup 1 > gpurun_out/r4_related_base_line.json 2> gpurun_out/r4_related_base_line.err || { tail -5 gpurun_out/r4_related_base_line.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_related_base_line.json").read().strip().splitlines()[-1])
print("related: %.3f M pairs/s, kernel %.1f ms per %d pairs, frac %.4f, cpu %.0f/s on %d cores, parity %s / %s" % (d["value"]/1e6, d["roofline"]["avg_launch_ms"], d["config"]["pairs_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d.get("parity_on_last_slab"), d["cpu_baseline"]["parity_on_sample"]))
PY
LZANI_LIB=$PWD/build/exp/stamps.so timeout -k 10 300 python bench.py $REL --steps 2 --warmup 1 --cpu-sample 0 --no-check > gpurun_out/r4_related_stamps.json 2> gpurun_out/r4_related_stamps.log || { tail -5 gpurun_out/r4_related_stamps.log; exit 1; }
grep "lzani stamps" gpurun_out/r4_related_stamps.log | tail -2
bash tools/profile.sh r4_related_base $REL || exit 1
cat gpurun_out/prof_r4_related_base/pmc.csv | cut -d, -f2,5,6 | head -40
