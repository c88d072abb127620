#!/bin/bash
# GPU box, round 4 run 19: presence matrix on the hash's top bits where the genomes fill little of the key space (mid-size genomes, long k-mers)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "mid_size or from_index or natural_trigger or bacterial or config4 or presence or bitmap or lists or run_time or fuzz or long_kmers" > gpurun_out/r4_run19_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run19_pytest.log; exit 1; }
grep -E "passed|failed|mal 15|kbp" gpurun_out/r4_run19_pytest.log | tail -8
timeout -k 10 300 python tools/fuzz_gpu.py 881 120 medium > gpurun_out/r4_fuzz_medium19.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_medium19.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_medium19.log
timeout -k 10 300 python tools/fuzz_gpu.py 882 60 large > gpurun_out/r4_fuzz_large19.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_large19.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_large19.log
