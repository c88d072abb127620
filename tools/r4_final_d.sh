#!/bin/bash
# GPU box, end of round 4, part D: the differential fuzz of the final library (small / medium / large / run-time compiled tuples; medium also with the split forced)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r4_final_fuzz.log
run() { echo "== $*" >> gpurun_out/r4_final_fuzz.log; timeout -k 10 500 "$@" > gpurun_out/fz.tmp 2>&1 || { tail -20 gpurun_out/fz.tmp; exit 1; }; tail -1 gpurun_out/fz.tmp | tee -a gpurun_out/r4_final_fuzz.log; }
run python tools/fuzz_gpu.py 3101 100 || exit 1
run python tools/fuzz_gpu.py 3102 150 medium || exit 1
run python tools/fuzz_gpu.py 3103 150 large || exit 1
LZANI_SPLIT=1 LZANI_SPLIT_SEGLEN=2500 LZANI_PM_MIN_ROWS=1 run python tools/fuzz_gpu.py 3104 120 medium || exit 1
run python tools/fuzz_gpu.py 3105 150 rtc 10 || exit 1
