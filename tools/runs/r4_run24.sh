#!/bin/bash
# GPU box, round 4 run 24: where the split's time goes at 32 x 5 Mbp: rounds timed; every pair whole through the segment kernel
set -o pipefail
mkdir -p gpurun_out
B="--genomes 32 --lmin 4500000 --lmax 5500000 --seed 3 --params mal=15,msl=9,reg=60 --slab 32 --steps 1 --warmup 1 --cpu-sample 0 --no-check"
LZANI_TRACE=1 timeout -k 10 600 python bench.py $B > gpurun_out/r24_a.json 2> gpurun_out/r24_a.err; grep "split:" gpurun_out/r24_a.err | tail -6
echo "--- no pair cut (threshold beyond every count): whole pairs through k_split"
LZANI_SPLIT_THR=4000000000 LZANI_TRACE=1 timeout -k 10 600 python bench.py $B > gpurun_out/r24_b.json 2> gpurun_out/r24_b.err; grep "split:" gpurun_out/r24_b.err | tail -3
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "split_over or natural_trigger" > gpurun_out/r4_run24_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run24_pytest.log; exit 1; }
grep -E "passed|failed|split:|5 Mbp" gpurun_out/r4_run24_pytest.log | tail -8
