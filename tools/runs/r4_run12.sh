#!/bin/bash
# GPU box, round 4 run 12: the group path with persistent buffers and the staged copy-out: its tests, then the 10k end-to-end run of the host binary
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "group or host_cli or tiled or dist or rccl" > gpurun_out/r4_run12_pytest.log 2>&1 || { tail -40 gpurun_out/r4_run12_pytest.log; exit 1; }
tail -2 gpurun_out/r4_run12_pytest.log
timeout -k 10 600 bash tools/e2e_cli.sh 10000 2 > gpurun_out/r4_e2e_10k.log 2>&1 || { tail -20 gpurun_out/r4_e2e_10k.log; exit 1; }
tail -12 gpurun_out/r4_e2e_10k.log
