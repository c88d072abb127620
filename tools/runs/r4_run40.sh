#!/bin/bash
# GPU box, round 4 run 40 (the last GPU minutes): the split from 8 wave slots a pair on: its tests, a short forced-split fuzz, 32 x 5 Mbp by itself, 128 x 5 Mbp forced
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -k "split_over or natural_trigger" > gpurun_out/r4_run40_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run40_pytest.log; exit 1; }
tail -1 gpurun_out/r4_run40_pytest.log
LZANI_SPLIT=1 LZANI_SPLIT_SEGLEN=2500 LZANI_PM_MIN_ROWS=1 timeout -k 10 120 python tools/fuzz_gpu.py 5001 40 medium > gpurun_out/r4_fuzz_medium40.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_medium40.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_medium40.log
bash tools/c4_bench.sh 32 | tee gpurun_out/r4_final_c4_32.txt
bash tools/c4_bench.sh 128 LZANI_SPLIT=1 | tee gpurun_out/r4_c4_128_forced.txt
