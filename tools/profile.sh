#!/bin/bash
# GPU box: rocprofv3 evidence for the bench workload.  Usage (through gpurun):  bash tools/profile.sh <tag> [bench.py arguments of another workload]
#   1. --kernel-trace --stats of the default bench command           -> gpurun_out/prof_<tag>/stats
#   2. --pmc passes (own runs, one k_pairs launch each, no other trace domain) -> gpurun_out/prof_<tag>/pmc_*
#   3. tools/profile_summary.py folds them into gpurun_out/prof_<tag>/{kernel_stats.csv,pmc.csv,pmc.json}
# Copy the three summary files into profiles/ afterwards.
set -o pipefail
TAG=${1:-run}
shift
EXTRA="$@"                      # e.g. --genomes 128 --lmin 4500000 --lmax 5500000 --seed 3 --params mal=15,msl=9,reg=60 --slab 128
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py"
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$B" --steps 4 --warmup 1 --cpu-sample 0 $EXTRA > "$OUT/stats.log" 2>&1 || { tail -5 "$OUT/stats.log"; exit 1; }
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"; do
    name=$(echo "$set" | tr ' ' '_' | cut -c1-40)
    timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d "$OUT/pmc_$name" -- python3 "$B" --steps 1 --warmup 0 --cpu-sample 0 $EXTRA > "$OUT/pmc_$name.log" 2>&1 || { echo "pmc pass '$set' failed"; tail -3 "$OUT/pmc_$name.log"; }
done
cd "$ROOT" && python3 tools/profile_summary.py "$OUT"
