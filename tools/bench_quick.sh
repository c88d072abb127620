#!/bin/bash
# quick GPU loop: parity tests, then the bench line (value / launch ms / roofline frac)
set -o pipefail
timeout -k 10 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 400 python bench.py --steps 3 --warmup 1 "$@" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.readline())
r=d['roofline']; c=d.get('cpu_baseline') or {}
print('pairs/s %.0f  launch_ms %.1f  frac %.4f  index_ms %.1f  cpu %.0f (%s cores) parity %s' % (d['value'], r['avg_launch_ms'], r['frac'], r['index_build_ms_per_step'], c.get('value',0), c.get('cores'), c.get('parity_on_sample')))"
