#!/bin/bash
# GPU box, round 4 run 26: split with later checkpoints: parity; 8 x 5 Mbp by the join form (default below 32 rows), by bitmaps unsplit / split
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "split_over or natural_trigger or bacterial" > gpurun_out/r4_run26_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run26_pytest.log; exit 1; }
grep -E "passed|failed|split:|5 Mbp" gpurun_out/r4_run26_pytest.log | tail -8
bash tools/c4_bench.sh 8 || exit 1
bash tools/c4_bench.sh 8 LZANI_PM_MIN_ROWS=1 LZANI_SPLIT=0 || exit 1
bash tools/c4_bench.sh 8 LZANI_PM_MIN_ROWS=1 || exit 1
bash tools/c4_bench.sh 16 || exit 1
bash tools/c4_bench.sh 16 LZANI_PM_MIN_ROWS=1 || exit 1
bash tools/c4_bench.sh 32 || exit 1
