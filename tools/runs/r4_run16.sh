#!/bin/bash
# GPU box, round 4 run 16: the own radix sort (numpy parity, index-build parity), filtered mid-size rows at mal 15, then 128 x 5 Mbp
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "radix_sort or index_build or last_slot or mid_size or from_index or natural_trigger or bacterial or config4 or lists" > gpurun_out/r4_run16_pytest.log 2>&1 || { tail -40 gpurun_out/r4_run16_pytest.log; exit 1; }
grep -E "passed|failed|mal 15" gpurun_out/r4_run16_pytest.log | tail -6
bash tools/c4_bench.sh 128 || exit 1
bash tools/c4_bench.sh 128 LZANI_NO_JOIN=0 LZANI_PM=0 || exit 1
