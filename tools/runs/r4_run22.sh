#!/bin/bash
# GPU box, round 4 run 22: one pair by several waves (lzani_kernels_split.h): forced at small sizes against the oracle, by itself at 33 x 5 Mbp, then the 32 / 128 x 5 Mbp rates
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "split_over" > gpurun_out/r4_run22_pytest_a.log 2>&1 || { tail -30 gpurun_out/r4_run22_pytest_a.log; exit 1; }
grep -E "passed|failed|split:" gpurun_out/r4_run22_pytest_a.log | tail -8
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "natural_trigger or bacterial or config4" > gpurun_out/r4_run22_pytest_b.log 2>&1 || { tail -30 gpurun_out/r4_run22_pytest_b.log; exit 1; }
grep -E "passed|failed|5 Mbp" gpurun_out/r4_run22_pytest_b.log | tail -8
bash tools/c4_bench.sh 32 LZANI_SPLIT=0 || exit 1
bash tools/c4_bench.sh 32 || exit 1
bash tools/c4_bench.sh 128 || exit 1
