#!/bin/bash
# GPU box, round 4 run 35: the matrix-from-index test with 64 KB chunks; the radix sort's tile (8 / 16 / 24 keys a thread) at 128 x 5 Mbp
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "from_index or natural_trigger or radix or index_build or bacterial" > gpurun_out/r4_run35_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run35_pytest.log; exit 1; }
tail -1 gpurun_out/r4_run35_pytest.log
bash tools/c4_bench.sh 128
bash tools/c4_bench.sh 128 LZANI_LIB=$PWD/build/exp/rs8.so
bash tools/c4_bench.sh 128 LZANI_LIB=$PWD/build/exp/rs24.so
