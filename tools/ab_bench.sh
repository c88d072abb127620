#!/bin/bash
# GPU box: the 10k bench for the shipped library and for each diagnostic build given (A/B of kernel experiments).
# Usage: tools/ab_bench.sh build/exp/a.so build/exp/b.so ...   (steps 10, warmup 3, no CPU baseline)
mkdir -p gpurun_out
for lib in "" "$@"; do
    name=${lib:-shipped}
    LZANI_LIB=${lib:+$PWD/$lib} timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 > gpurun_out/ab_$(basename "$name" .so).json 2>gpurun_out/ab_err.log || { echo "FAILED $name"; tail -5 gpurun_out/ab_err.log; exit 1; }
    python - "$name" gpurun_out/ab_$(basename "$name" .so).json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("%-28s %.3f M pairs/s  %.1f ms/step  parity %s" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"], d.get("parity_on_last_slab")))
PY
done
