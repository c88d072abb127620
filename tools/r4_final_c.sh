#!/bin/bash
# GPU box, end of round 4, part C: the whole -m gpu suite and the bench lines again on the final library (after the split work)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r4_final_pytest.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/r4_final_pytest.log | tail -20; exit 1; }
tail -1 gpurun_out/r4_final_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r4_final_bench_default_line.json 2> gpurun_out/r4_final_bench.err || { tail -5 gpurun_out/r4_final_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_final_bench_default_line.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("default bench: %.3f M pairs/s, %.1f ms/step, kernel %.1f ms, cand %.1f ms, frac %.4f, cpu %.0f pairs/s, parity %s / %s" % (d["value"]/1e6, d["ms_per_step"], r["avg_launch_ms"], r["candidate_stage_ms_per_step"], r["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["parity_on_sample"], d["parity_on_last_slab"]))
PY
for D in 0.15 0.05; do
timeout -k 10 600 python bench.py --workload related --genomes 20000 --fam 50 --seed 1 --dmax $D --steps 4 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('related d<=$D: %.3f M pairs/s, kernel %.1f ms, parity %s' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['parity_on_last_slab']))"
done
bash tools/c4_bench.sh 128
bash tools/c4_bench.sh 32
