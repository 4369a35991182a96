#!/bin/bash
# usage: tools/kernel_resources.sh timeband_amd/csrc/lstm.hip  -> VGPR / AGPR / scratch per kernel
cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -c "$OLDPWD/$1" -o /tmp/_res_chk.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
name=None;rows={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); rows[name]={}
    for k in ('VGPRs:','AGPRs','ScratchSize','Occupancy'):
        if k in l and name: rows[name][k.strip(':')]=l.split(':')[-1].split('[')[0].strip()
for n,r in rows.items():
    m=re.search(r'(\w+?_kernel)(ILi[\dELi]+E)?',n)
    t=re.findall(r'Li(\d+)',n)
    print((m.group(1) if m else n)[:24].ljust(24), ','.join(t).ljust(14), ' '.join(f'{k}={v}' for k,v in r.items()))
"
