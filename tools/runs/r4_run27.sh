#!/bin/bash
# GPU box, round 4 run 27: the whole -m gpu suite with the split in its default place (few long pairs), the fuzz (large: now bitmaps + split), rates at 8 / 16 x 5 Mbp
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r4_run27_pytest.log 2>&1 || { grep -E "^(FAILED|ERROR)|passed|failed|Error" gpurun_out/r4_run27_pytest.log | tail -20; exit 1; }
tail -1 gpurun_out/r4_run27_pytest.log
timeout -k 10 400 python tools/fuzz_gpu.py 991 150 large > gpurun_out/r4_fuzz_large27.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_large27.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_large27.log
timeout -k 10 300 python tools/fuzz_gpu.py 992 60 medium > gpurun_out/r4_fuzz_medium27.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_medium27.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_medium27.log
bash tools/c4_bench.sh 8 | tee gpurun_out/r4_final_c4_8.txt
bash tools/c4_bench.sh 16 | tee gpurun_out/r4_final_c4_16.txt
