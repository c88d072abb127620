#!/bin/bash
# Static instruction counts of k_pairs per section (markers at the stamp() sites).  Usage: tools/isa_sections.sh [kernel-mangled-substring]
K=${1:-k_pairsILb1ELb1ELb1}
cd /root/repo/lz-ani_amd
python3 - <<'PY'
import re
s=open('csrc/lzani_hip.hip').read()
s=s.replace('"../../include/lzani.h"','"/root/repo/include/lzani.h"')
open('/tmp/mark.hip','w').write(s)
k=open('csrc/lzani_kernels_pairs.h').read()
k=re.sub(r'(?<![\w.])stamp\((\d)\);', lambda m: 'asm volatile("; LZMARK %s");' % m.group(1), k)
open('/tmp/lzani_kernels_pairs.h','w').write(k)
c=open('csrc/lzani_core.h').read()
c=re.sub(r'\bw\.stamp\((\d)\);', lambda m: 'LZMARK(%s);' % m.group(1), c)
c=c.replace('namespace lzani {\n\ntypedef uint64_t u64;','#if defined(__HIP_DEVICE_COMPILE__)\n#define LZMARK(k) asm volatile("; LZMARK " #k)\n#else\n#define LZMARK(k)\n#endif\nnamespace lzani {\n\ntypedef uint64_t u64;',1)
open('/tmp/lzani_core.h','w').write(c)
PY
cp csrc/lzani_layout.h csrc/lzani_kernels_index.h csrc/lzani_kernels_cand.h csrc/lzani_multi.h /tmp/
cd /tmp && rm -f mark-hip-* && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-value -save-temps -o /tmp/mark.so /tmp/mark.hip -lrccl 2>/dev/null
S=$(ls /tmp/mark-hip-amdgcn*gfx950*.s | head -1)
awk "/^_ZN5lzani7${K}.*:/,/\.end_amdhsa_kernel/" $S > /tmp/kmark.s
python3 - <<'PY'
lines=open('/tmp/kmark.s').read().split('\n')
marks=[(i,l.strip()) for i,l in enumerate(lines) if 'LZMARK' in l]
def count(a,b):
    v=s=m=l=br=rl=0
    for x in lines[a:b]:
        t=x.strip()
        if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'): continue
        op=t.split()[0]
        if op.startswith('v_readlane') or op.startswith('v_writelane'): rl+=1; v+=1
        elif op.startswith('v_'): v+=1
        elif op.startswith('s_cbranch') or op.startswith('s_branch'): br+=1; s+=1
        elif op.startswith('s_'): s+=1
        elif op.startswith('global_') or op.startswith('scratch_') or op.startswith('flat_'): m+=1
        elif op.startswith('ds_'): l+=1
    return dict(VALU=v,SALU=s,VMEM=m,LDS=l,branches=br,spill_lane_ops=rl)
bounds=[0]+[i for i,_ in marks]+[len(lines)]
labels=['prologue']+[m for _,m in marks]
for k in range(len(bounds)-1):
    print(f'{labels[k]:12s} lines {bounds[k]:5d}-{bounds[k+1]:5d}', count(bounds[k],bounds[k+1]))
print('total lines', len(lines))
PY
