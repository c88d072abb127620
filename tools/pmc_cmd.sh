#!/bin/bash
# GPU box: one rocprofv3 --pmc pass of an arbitrary python command, k_pairs counters summed.  Usage: tools/pmc_cmd.sh "<counters>" script.py [args]
set -o pipefail
CNT=$1; shift
ROOT=$(pwd)
D=$ROOT/gpurun_out/pmcc_$$
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc $CNT --output-format csv -d "$D" -- python3 "$ROOT/$1" "${@:2}" > "$D.log" 2>&1 || { tail -3 "$D.log"; exit 1; }
cd "$ROOT" && python3 - "$D" <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pairs" in r["Kernel_Name"]:
            a = acc.setdefault(r["Counter_Name"], [0.0, set()])
            a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
print({k: "%.4g per launch" % (v[0] / max(1, len(v[1]))) for k, v in sorted(acc.items())})
PY
rm -rf "$D"
