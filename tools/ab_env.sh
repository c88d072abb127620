#!/bin/bash
# GPU box: the 10k bench (steps 10, warmup 3, no CPU baseline) for a list of "name:lib:ENV=V,ENV=V" variants.
# Usage: tools/ab_env.sh shipped:: occ7:build/exp/occ7.so:LZANI_BLOCKS_PER_CU=7
mkdir -p gpurun_out
for spec in "$@"; do
    IFS=: read -r name lib envs <<< "$spec"
    (
        [ -n "$lib" ] && export LZANI_LIB=$PWD/$lib
        IFS=, read -ra kv <<< "$envs"
        for e in "${kv[@]}"; do [ -n "$e" ] && export "$e"; done
        timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || { echo "FAILED $name"; tail -5 gpurun_out/ab_$name.err; exit 1; }
    ) || exit 1
    python - "$name" gpurun_out/ab_$name.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("%-20s %.3f M pairs/s  %.1f ms/step  kernel %.1f ms  parity %s" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d.get("parity_on_last_slab")))
PY
done
