#!/bin/bash
# GPU box: BASELINE configs[3] shape on one GPU (N x 5 Mbp, --mal 15 --msl 9 --reg 60).  Usage: tools/c4_bench.sh [n_genomes] [env assignments...]
N=${1:-128}; shift
for kv in "$@"; do export "$kv"; done
timeout -k 10 800 python bench.py --genomes $N --lmin 4500000 --lmax 5500000 --seed 3 --params mal=15,msl=9,reg=60 --slab $N --steps 2 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('genomes', d['config']['genomes'], 'pairs/s %.0f' % d['value'], 'ms/step %.0f' % d['ms_per_step'], 'k_pairs ms %.0f' % r['avg_launch_ms'], 'index ms %.0f' % r['index_build_ms_per_step'], 'cand ms %.0f' % r['candidate_stage_ms_per_step'], 'frac %.4f' % r['frac'], d.get('parity_on_last_slab'), 'NO_JOIN=' + os.environ.get('LZANI_NO_JOIN', '0'))"
