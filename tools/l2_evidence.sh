#!/bin/bash
# GPU box: the evidence behind "the viral pair kernel is bound by the L2 -> L1 line traffic of the tag-word probes"
# (DESIGN.md section 6): the 10k bench with every probe made twice (diagnostic build -DLZANI_PROBE2X, given as $1), for
# the wave kernel and for the block kernel with the LDS filter, and the L1 -> L2 request counters of both.
P2X=$1
line() { python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('%-58s %.3f M pairs/s  %.1f ms/step  kernel %s %.1f ms  parity %s' % (sys.argv[1], d['value']/1e6, d['ms_per_step'], r['kernel'], r['avg_launch_ms'], d.get('parity_on_last_slab')))" "$1"; }
B="python bench.py --steps 8 --warmup 2 --cpu-sample 0"
LZANI_BLOCK_KERNEL=0 $B 2>/dev/null | line "wave kernel (no filter)"
LZANI_BLOCK_KERNEL=0 LZANI_LIB=$PWD/$P2X $B 2>/dev/null | line "wave kernel, every probe twice"
$B 2>/dev/null | line "block kernel, filter of 2^18 bits in LDS"
LZANI_LIB=$PWD/$P2X $B 2>/dev/null | line "block kernel, every (unfiltered) probe twice"
LZANI_FILTER_MAX_BITS=17 $B 2>/dev/null | line "block kernel, filter of 2^17 bits"
LZANI_FILTER_MAX_BITS=16 $B 2>/dev/null | line "block kernel, filter of 2^16 bits"
C="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"
echo "counters per launch of 4,999,500 pairs, wave kernel:"; LZANI_BLOCK_KERNEL=0 bash tools/pmc_quick.sh "$C"
echo "counters per launch of 4,999,500 pairs, block kernel:"; bash tools/pmc_quick.sh "$C"
