#!/bin/bash
# GPU box, round 4 run 18: the mid-size filtered test at its new size; kernel stats of 128 x 5 Mbp (where the index stage's 68 ms go)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "mid_size" > gpurun_out/r4_run18_pytest.log 2>&1 || { tail -30 gpurun_out/r4_run18_pytest.log; exit 1; }
grep -E "passed|failed|mal 15" gpurun_out/r4_run18_pytest.log | tail -4
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_c4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --genomes 128 --lmin 4500000 --lmax 5500000 --seed 3 --params mal=15,msl=9,reg=60 --slab 128 --steps 4 --warmup 1 --cpu-sample 0 > "$OUT/stats.log" 2>&1 || { tail -5 "$OUT/stats.log"; exit 1; }
cd $ROOT
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r4_c4_128x5mbp_kernel_stats.csv
cut -c1-150 $f | head -24
grep '^{' $OUT/stats.log | tail -1 > gpurun_out/r4_c4_128x5mbp_bench_line.json
