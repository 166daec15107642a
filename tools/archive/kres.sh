#!/bin/bash
# usage: tools/kres.sh <file.hip>  -- per-kernel register / scratch / occupancy summary (hipcc remarks)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'remark: [^ ]+ +([\w \[\]/]+): (\S+)',l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k.endswith('Name'):
        if cur: print(cur)
        cur=v[:70]+' |'
    elif k.split()[0] in ('VGPRs','ScratchSize','Occupancy','LDS','TotalSGPRs') or 'Spill' in k:
        cur+=' %s=%s'%(k.split(' [')[0].replace(' ',''),v)
if cur: print(cur)
"
