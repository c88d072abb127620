#!/bin/bash
# GPU box: k_idx_build duration for the diagnostic variants (results are wrong in these builds; timing only)
cd /tmp && export TMPDIR=/tmp
for f in "" $GRAFT_REPO_ROOT/tools/diag/v_*.so; do
  rm -rf /tmp/pi; LZANI_LIB=$f timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -- python3 $GRAFT_REPO_ROOT/tools/diag/idxrun.py > /dev/null 2>&1
  g=$(find /tmp/pi -name "*kernel_stats.csv" | head -1)
  echo "variant '$f': $(grep k_idx_build $g | cut -d, -f1-4 | cut -c1-120)"
done
