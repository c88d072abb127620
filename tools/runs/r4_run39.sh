#!/bin/bash
# GPU box, round 4 run 39: checkpoints searched until the open region spans reg positions: parity of the split tests, rates at 8 / 16 / 32 (forced) x 5 Mbp
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "split_over or bacterial" > gpurun_out/r4_run39_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run39_pytest.log; exit 1; }
tail -1 gpurun_out/r4_run39_pytest.log
bash tools/c4_bench.sh 8 | tee gpurun_out/r4_final_c4_8.txt
bash tools/c4_bench.sh 16 | tee gpurun_out/r4_final_c4_16.txt
bash tools/c4_bench.sh 32 LZANI_SPLIT=1 | tee gpurun_out/r4_c4_32_forced.txt
