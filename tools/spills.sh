#!/bin/bash
# where the spills of map_kernel<64,false> sit: source line + loop depth of every scratch access.  tools/spills.sh [-D...]
ROOT=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-align-mismatch -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -pragma-unroll-threshold=65536 -gline-tables-only "$@" -S --cuda-device-only -o /tmp/spills.s $ROOT/mapquik_amd/csrc/mq_capi.hip 2>/dev/null
python3 - <<'PY'
import re,os
L=open('/tmp/spills.s').read().split('\n')
files={}
k=None; loc=None; depth=0; out={}
for l in L:
    m=re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?',l)
    if m: files[m.group(1)]=(m.group(3) or m.group(2)).split('/')[-1]
    if re.match(r'^_Z\w+:',l): k=l.split(':')[0]
    m=re.match(r'\s*\.loc\s+(\d+)\s+(\d+)',l)
    if m: loc=(files.get(m.group(1),m.group(1)),int(m.group(2)))
    m=re.search(r'Depth=(\d+)',l)
    if m and l.startswith('.LBB'): depth=int(m.group(1))
    elif l.startswith('.LBB'): depth=0
    if 'scratch_' in l and k and os.environ.get("SPILL_KERNEL","map_kernelILi64ELb0ELb0") in k:
        kind='store' if 'store' in l else 'load'
        key=(loc,depth,kind)
        out[key]=out.get(key,0)+1
for (loc,d,kind),n in sorted(out.items(), key=lambda x:(-x[0][1], str(x[0][0]))):
    print("depth %d  %-5s x%d  %s:%s" % (d,kind,n,loc[0] if loc else '?',loc[1] if loc else '?'))
PY
