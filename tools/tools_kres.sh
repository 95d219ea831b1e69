#!/bin/bash
# compile the three units of libpyrovi.so and print a compact per-kernel resource table (optionally filtered by $1)
(for u in pyrovi f64 lean; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -c -o /tmp/kres_$u.o pyro_amd/csrc/$u.hip -Rpass-analysis=kernel-resource-usage; done) 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for l in sys.stdin:
    if 'error' in l or 'warning:' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ('TotalSGPRs','VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','LDS Size \[bytes/block\]'):
        m=re.search(k+r': (\d+)',l)
        if m and cur: rows[cur][k[:6]]=m.group(1)
flt=sys.argv[1] if len(sys.argv)>1 else ''
for k,v in rows.items():
    if flt in k: print('%-70s'%k[:70], v)
" "$1"
