#!/bin/bash
# GPU box, end of round 4, part B: related workload (lines + rocprofv3 evidence), the long-genome configuration (128 / 32 x 5 Mbp, the full 1,000 x 5 Mbp job), a run-time compiled tuple
set -o pipefail
mkdir -p gpurun_out
REL="--workload related --genomes 20000 --fam 50 --seed 1"
for D in 0.15 0.05; do
timeout -k 10 600 python bench.py $REL --dmax $D --steps 4 --warmup 1 > gpurun_out/r4_final_related_line_d$D.json 2> gpurun_out/r4_final_related_$D.err || { tail -5 gpurun_out/r4_final_related_$D.err; exit 1; }
python - $D <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r4_final_related_line_d%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
cb = d.get("cpu_baseline") or {}
print("related d<=%s: %.3f M pairs/s, kernel %.1f ms, index %.1f ms per %d pairs, frac %.4f, cpu %s pairs/s, parity %s" % (sys.argv[1], d["value"]/1e6, r["avg_launch_ms"], r["index_build_ms_per_step"], d["config"]["pairs_per_step"], r["frac"], cb.get("value"), d.get("parity_on_last_slab")))
PY
done
bash tools/profile.sh r4_related_final $REL --dmax 0.15 || exit 1
bash tools/c4_bench.sh 128 | tee gpurun_out/r4_final_c4_128.txt
bash tools/c4_bench.sh 32 | tee gpurun_out/r4_final_c4_32.txt
timeout -k 10 300 python bench.py --params reg=36 --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r4_final_rtc_reg36_line.json 2> gpurun_out/r4_final_rtc.err || { tail -5 gpurun_out/r4_final_rtc.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/r4_final_rtc_reg36_line.json').read().strip().splitlines()[-1]); print('reg=36: %.3f M pairs/s, kernel %.1f ms, parity %s' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['parity_on_last_slab']))"
timeout -k 10 500 bash tools/c4_full.sh 1000 3 200 > gpurun_out/r4_final_config4_full_1000x5mbp.log 2>&1; tail -12 gpurun_out/r4_final_config4_full_1000x5mbp.log
