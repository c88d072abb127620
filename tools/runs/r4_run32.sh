#!/bin/bash
# GPU box, round 4 run 32: k_pm_cand with 512 threads a block (16 waves a CU) against 256: parity of the candidate stage, then the default bench A/B and 128 x 5 Mbp
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "presence or bitmap or lists or from_index or fuzz or vir61 or reference_vectors or natural_trigger" > gpurun_out/r4_run32_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run32_pytest.log; exit 1; }
tail -1 gpurun_out/r4_run32_pytest.log
bash tools/ab_env.sh t512:: t256:build/exp/cand256.so: t512b::
bash tools/c4_bench.sh 128
bash tools/c4_bench.sh 128 LZANI_LIB=$PWD/build/exp/cand256.so
