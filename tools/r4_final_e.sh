#!/bin/bash
# GPU box, end of round 4, part E: rocprofv3 kernel stats of 128 x 5 Mbp on the final library
set -o pipefail
mkdir -p gpurun_out
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_c4; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --genomes 128 --lmin 4500000 --lmax 5500000 --seed 3 --params mal=15,msl=9,reg=60 --slab 128 --steps 4 --warmup 1 --cpu-sample 0 > "$OUT/stats.log" 2>&1 || { tail -5 "$OUT/stats.log"; exit 1; }
cd $ROOT
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r4_c4_128x5mbp_kernel_stats.csv
grep '^{' $OUT/stats.log | tail -1 > gpurun_out/r4_c4_128x5mbp_bench_line.json
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r4_c4_128x5mbp_kernel_stats.csv")))[:9]:
    print("%-70s calls %3s avg %9.3f ms total %8.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6))
PY
