#!/bin/bash
# GPU box, round 4 run 15: presence matrix from the indexes (k_pm_from_index) + longest pair first: parity, then 128 x 5 Mbp A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "from_index or natural_trigger or bacterial or config4 or presence or bitmap or lists" > gpurun_out/r4_run15_pytest.log 2>&1 || { tail -40 gpurun_out/r4_run15_pytest.log; exit 1; }
tail -3 gpurun_out/r4_run15_pytest.log
bash tools/c4_bench.sh 128 LZANI_PM_FROM_INDEX=0 LZANI_LPT=0 || exit 1
bash tools/c4_bench.sh 128 LZANI_LPT=0 || exit 1
bash tools/c4_bench.sh 128 || exit 1
bash tools/c4_bench.sh 32 LZANI_LPT=0 || exit 1
bash tools/c4_bench.sh 32
