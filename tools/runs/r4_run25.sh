#!/bin/bash
# GPU box, round 4 run 25: the split where pairs are VERY few (8 / 16 genomes x 5 Mbp): unsplit, heavy pairs cut, all pairs cut
set -o pipefail
mkdir -p gpurun_out
export LZANI_PM_MIN_ROWS=1
for N in 8 16; do
echo "--- $N genomes"
bash tools/c4_bench.sh $N LZANI_SPLIT=0 || exit 1
bash tools/c4_bench.sh $N LZANI_SPLIT=1 || exit 1
bash tools/c4_bench.sh $N LZANI_SPLIT=1 LZANI_SPLIT_ALL=1 || exit 1
done
B="--genomes 8 --lmin 4500000 --lmax 5500000 --seed 3 --params mal=15,msl=9,reg=60 --slab 8 --steps 1 --warmup 1 --cpu-sample 0 --no-check"
LZANI_SPLIT=1 LZANI_SPLIT_ALL=1 LZANI_TRACE=1 timeout -k 10 600 python bench.py $B > gpurun_out/r25_a.json 2> gpurun_out/r25_a.err; grep "split:" gpurun_out/r25_a.err | tail -8
