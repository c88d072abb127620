#!/bin/bash
# GPU box: bench value for every diagnostic library variant under tools/diag/v_*.so
for f in tools/diag/v_*.so; do
  v=$(LZANI_LIB=$f timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.0f %.1f' % (d['value'], d['roofline']['avg_launch_ms']))")
  echo "$f $v"
done
