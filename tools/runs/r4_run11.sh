#!/bin/bash
# GPU box, round 4 run 11: BASELINE configs[3] (1,000 x 5 Mbp) at full size, 200 sampled rows against the reference's parser; then 128 x 5 Mbp kernel stats
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 bash tools/c4_full.sh 1000 3 200 > gpurun_out/r4_config4_full_1000x5mbp.log 2>&1 || { tail -20 gpurun_out/r4_config4_full_1000x5mbp.log; exit 1; }
grep -E "engine:|GPU 0|LZ matching|Total time|sampled rows" gpurun_out/r4_config4_full_1000x5mbp.log
