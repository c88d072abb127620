#!/bin/bash
# GPU box: per-path event counters of the related workload (diagnostic build build/exp/paths.so, -DLZANI_PATH_STATS)
set -o pipefail
mkdir -p gpurun_out
for D in 0.15 0.05; do
LZANI_LIB=$PWD/build/exp/paths.so timeout -k 10 300 python bench.py --workload related --genomes 8000 --fam 50 --dmax $D --seed 1 --steps 1 --warmup 1 --cpu-sample 0 --no-check > gpurun_out/r4_paths_$D.json 2> gpurun_out/r4_paths_$D.log || { tail -5 gpurun_out/r4_paths_$D.log; exit 1; }
grep "lzani paths" gpurun_out/r4_paths_$D.log | tail -1 | tr ';' '\n'
done
