#!/bin/bash
# GPU box: one rocprofv3 --pmc pass of the bench workload, k_pairs counters summed.  Usage: tools/pmc_quick.sh "<counters>" [bench args]
set -o pipefail
CNT=$1; shift
ROOT=$(pwd)
D=$ROOT/gpurun_out/pmcq_$$
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $CNT --output-format csv -d "$D" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --cpu-sample 0 "$@" > "$D.log" 2>&1 || { tail -3 "$D.log"; exit 1; }
cd "$ROOT" && python3 - "$D" <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pairs" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
print({k: "%.4g" % v for k, v in sorted(acc.items())})
PY
rm -rf "$D"
