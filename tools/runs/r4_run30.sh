#!/bin/bash
# GPU box, round 4 run 30: void segments by cause at 8 x 5 Mbp
mkdir -p gpurun_out
B="--genomes 8 --lmin 4500000 --lmax 5500000 --seed 3 --params mal=15,msl=9,reg=60 --slab 8 --steps 1 --warmup 0 --cpu-sample 0 --no-check"
LZANI_TRACE=1 timeout -k 10 600 python bench.py $B > gpurun_out/r30_a.json 2> gpurun_out/r30_a.err; grep "split:" gpurun_out/r30_a.err | tail -6
