#!/bin/bash
# GPU box: wave-cycle shares per section of the pair machine on the related workload (diagnostic build, -DLZANI_STAMPS)
set -o pipefail
mkdir -p gpurun_out
for D in 0.15 0.05; do
LZANI_LIB=$PWD/build/exp/stamps.so timeout -k 10 300 python bench.py --workload related --genomes 8000 --fam 50 --dmax $D --seed 1 --steps 1 --warmup 1 --cpu-sample 0 --no-check > gpurun_out/r4_stamps_$D.json 2> gpurun_out/r4_stamps_$D.log || { tail -5 gpurun_out/r4_stamps_$D.log; exit 1; }
grep "lzani stamps" gpurun_out/r4_stamps_$D.log | tail -1
done
