#!/bin/bash
# GPU box, round 4 run 29: the null chain inside segments whose look-back is a lower bound (record answers only): parity (forced split, long genomes, fuzz), rates
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -k "split_over or natural_trigger or bacterial or config4 or mid_size" > gpurun_out/r4_run29_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run29_pytest.log; exit 1; }
grep -E "passed|failed" gpurun_out/r4_run29_pytest.log | tail -3
timeout -k 10 400 python tools/fuzz_gpu.py 1991 120 large > gpurun_out/r4_fuzz_large29.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_large29.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_large29.log
LZANI_SPLIT=1 LZANI_SPLIT_SEGLEN=3000 LZANI_PM_MIN_ROWS=1 timeout -k 10 400 python tools/fuzz_gpu.py 1992 120 medium > gpurun_out/r4_fuzz_medium29.log 2>&1 || { tail -20 gpurun_out/r4_fuzz_medium29.log; exit 1; }
tail -1 gpurun_out/r4_fuzz_medium29.log
bash tools/c4_bench.sh 8 || exit 1
bash tools/c4_bench.sh 16 || exit 1
bash tools/c4_bench.sh 32 LZANI_SPLIT=1 || exit 1
B="--genomes 8 --lmin 4500000 --lmax 5500000 --seed 3 --params mal=15,msl=9,reg=60 --slab 8 --steps 1 --warmup 1 --cpu-sample 0 --no-check"
LZANI_TRACE=1 timeout -k 10 600 python bench.py $B > gpurun_out/r29_a.json 2> gpurun_out/r29_a.err; grep "split:" gpurun_out/r29_a.err | tail -7
