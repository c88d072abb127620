#!/bin/bash
# GPU box, round 4 run 36: the related workload with fewer waves a CU (fewer references in flight per XCD: do their tag words then share the L2?)
mkdir -p gpurun_out
REL="--workload related --genomes 20000 --fam 50 --seed 1 --dmax 0.15"
for B in 8 6 5 4; do
LZANI_BLOCKS_PER_CU=$B timeout -k 10 600 python bench.py $REL --steps 3 --warmup 1 --cpu-sample 0 --no-check 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('blocks per CU $B: related d<=0.15: %.3f M pairs/s, kernel %.1f ms' % (d['value']/1e6, d['roofline']['avg_launch_ms']))"
done
