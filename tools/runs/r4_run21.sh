#!/bin/bash
# GPU box, round 4 run 21: the full 1,000 x 5 Mbp job again (where do 2.6 s of host time inside the matching come from?), with and without the round-4 pieces
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "natural_trigger or bacterial or config4" > gpurun_out/r4_run21_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run21_pytest.log; exit 1; }
tail -1 gpurun_out/r4_run21_pytest.log
LZANI_TRACE=1 timeout -k 10 400 bash tools/c4_full.sh 1000 3 40 > gpurun_out/r4_run21_full_a.log 2>&1; grep -E "engine:|GPU 0|LZ matching|differing|batch 0|batch 1 |slots" gpurun_out/r4_run21_full_a.log | cut -c1-200 | head -12
LZANI_LPT=0 LZANI_PM_FROM_INDEX=0 timeout -k 10 400 bash tools/c4_full.sh 1000 3 40 > gpurun_out/r4_run21_full_b.log 2>&1; grep -E "engine:|GPU 0|LZ matching|differing" gpurun_out/r4_run21_full_b.log | cut -c1-200
bash tools/c4_bench.sh 128
