#!/bin/bash
# GPU box, round 4 run 28: the split forced where it is off by default (128 x 5 Mbp, 2 segments a pair; 32 x 5 Mbp, 8): the bench's own check of 2,000 pairs against the reference
set -o pipefail
mkdir -p gpurun_out
bash tools/c4_bench.sh 128 LZANI_SPLIT=1 || exit 1
bash tools/c4_bench.sh 32 LZANI_SPLIT=1 || exit 1
LZANI_SPLIT=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "natural_trigger or mid_size or config4" > gpurun_out/r4_run28_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run28_pytest.log; exit 1; }
grep -E "passed|failed" gpurun_out/r4_run28_pytest.log | tail -3
