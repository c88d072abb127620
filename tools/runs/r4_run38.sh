#!/bin/bash
# GPU box, round 4 run 38: 64 segments a pair as the cap: the split's tests, the large fuzz, 8 / 16 x 5 Mbp, the CLI on 12 x 5 Mbp with every row checked
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "split_over or natural_trigger or bacterial or config4" > gpurun_out/r4_run38_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run38_pytest.log; exit 1; }
tail -1 gpurun_out/r4_run38_pytest.log
timeout -k 10 300 python tools/fuzz_gpu.py 4001 100 large > gpurun_out/r4_fuzz_large38.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_large38.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_large38.log
bash tools/c4_bench.sh 8 | tee gpurun_out/r4_final_c4_8.txt
bash tools/c4_bench.sh 16 | tee gpurun_out/r4_final_c4_16.txt
timeout -k 10 600 bash tools/c4_full.sh 12 3 200 > gpurun_out/r4_final_c4_12_cli.log 2>&1; grep -E "GPU 0|checked" gpurun_out/r4_final_c4_12_cli.log
