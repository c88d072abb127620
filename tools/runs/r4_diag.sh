#!/bin/bash
# GPU box: path counters and section shares of the related workload (diagnostic builds)
set -o pipefail
mkdir -p gpurun_out
bash tools/runs/r4_paths.sh || exit 1
bash tools/runs/r4_stamps.sh
