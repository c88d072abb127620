#!/bin/bash
# GPU box, round 4 run 9: related-pair profile of the current build, then BASELINE configs[4] (100k filtered) and configs[3] (1000 x 5 Mbp) at full size
set -o pipefail
mkdir -p gpurun_out
bash tools/profile.sh r4_related_final --workload related --genomes 20000 --fam 50 --dmax 0.15 --seed 1 || exit 1
timeout -k 10 500 bash tools/c5_full.sh 100000 4 > gpurun_out/r4_config5_full_100k.log 2>&1 || { tail -20 gpurun_out/r4_config5_full_100k.log; exit 1; }
tail -6 gpurun_out/r4_config5_full_100k.log
