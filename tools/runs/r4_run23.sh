#!/bin/bash
# GPU box, round 4 run 23: the split with only the heavy pairs cut: parity, then 32 x 5 Mbp with the rounds traced
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "split_over or natural_trigger" > gpurun_out/r4_run23_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run23_pytest.log; exit 1; }
grep -E "passed|failed|split:|5 Mbp" gpurun_out/r4_run23_pytest.log | tail -8
LZANI_TRACE=1 timeout -k 10 600 python bench.py --genomes 32 --lmin 4500000 --lmax 5500000 --seed 3 --params mal=15,msl=9,reg=60 --slab 32 --steps 2 --warmup 1 --cpu-sample 0 > gpurun_out/r23_c4_32.json 2> gpurun_out/r23_c4_32.err; grep "split:" gpurun_out/r23_c4_32.err | tail -14
bash tools/c4_bench.sh 32 || exit 1
bash tools/c4_bench.sh 32 LZANI_SPLIT_ALL=1 || exit 1
