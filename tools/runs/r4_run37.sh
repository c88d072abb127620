#!/bin/bash
# GPU box, round 4 run 37: more segments a pair where pairs are very few (8 / 16 x 5 Mbp): 32 (default) / 64, rounds allowed in proportion; parity of the split tests
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "split_over or natural_trigger or bacterial" > gpurun_out/r4_run37_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run37_pytest.log; exit 1; }
tail -1 gpurun_out/r4_run37_pytest.log
bash tools/c4_bench.sh 8
bash tools/c4_bench.sh 16
bash tools/c4_bench.sh 8 LZANI_SPLIT_S=64
B="--genomes 8 --lmin 4500000 --lmax 5500000 --seed 3 --params mal=15,msl=9,reg=60 --slab 8 --steps 1 --warmup 0 --cpu-sample 0 --no-check"
LZANI_SPLIT_S=64 LZANI_TRACE=1 timeout -k 10 600 python bench.py $B > gpurun_out/r37_a.json 2> gpurun_out/r37_a.err; grep "split: round" gpurun_out/r37_a.err | tail -12 | cut -c1-120
