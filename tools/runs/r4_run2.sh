#!/bin/bash
# GPU box, round 4 run 2: run-time compiled pair kernels -- parity test, bench line at reg=36 and am=6, fuzz
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "run_time_compiled" > gpurun_out/r4_run2_pytest.log 2>&1 || { tail -40 gpurun_out/r4_run2_pytest.log; exit 1; }
tail -2 gpurun_out/r4_run2_pytest.log
for P in reg=36 am=6 mal=12,msl=8,reg=40; do
timeout -k 10 300 python bench.py --params $P --steps 5 --warmup 2 --cpu-sample 0 > gpurun_out/r4_rtc_${P//[=,]/_}_line.json 2> gpurun_out/r4_rtc_${P//[=,]/_}.err || { tail -5 gpurun_out/r4_rtc_${P//[=,]/_}.err; exit 1; }
python - "$P" gpurun_out/r4_rtc_${P//[=,]/_}_line.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%s (run-time compiled): %.3f M pairs/s, kernel %.1f ms, parity %s" % (sys.argv[1], d["value"]/1e6, d["roofline"]["avg_launch_ms"], d.get("parity_on_last_slab")))
PY
done
timeout -k 10 500 python tools/fuzz_gpu.py 4242 300 rtc 25 > gpurun_out/r4_rtc_fuzz.log 2>&1 || { tail -20 gpurun_out/r4_rtc_fuzz.log; exit 1; }
tail -3 gpurun_out/r4_rtc_fuzz.log
