#!/bin/bash
# GPU box, round 4 run 34: more waves a CU for two more latency-bound helper kernels: k_pm_from_index with 64 KB chunks (two blocks a CU), k_idx_build with 64 KB of LDS
set -o pipefail
mkdir -p gpurun_out
bash tools/c4_bench.sh 128
bash tools/c4_bench.sh 128 LZANI_PMFI_RCL=12
bash tools/c4_bench.sh 128 LZANI_PMFI_RCL=11
REL="--workload related --genomes 20000 --fam 50 --seed 1 --dmax 0.15"
for LIB in "" build/exp/idx64k.so; do
LZANI_LIB=${LIB:+$PWD/$LIB} timeout -k 10 600 python bench.py $REL --steps 4 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('${LIB:-shipped} related d<=0.15: %.3f M pairs/s, kernel %.1f ms, index %.1f ms, parity %s' % (d['value']/1e6, d['roofline']['avg_launch_ms'], d['roofline']['index_build_ms_per_step'], d['parity_on_last_slab']))"
done
