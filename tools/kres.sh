#!/bin/bash
# per-kernel register/occupancy table of one HIP source:  tools/kres.sh vspbfr_amd/csrc/conv_igemm.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)', l)
    if m: cur={'name':re.sub(r'_ZN12_GLOBAL__N_1\d+','',m.group(1))[:60]}; rows.append(cur); continue
    for k in ('VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','SGPRs Spill','TotalSGPRs'):
        m=re.search(k+r': (\d+)', l)
        if m and cur is not None: cur[k.split(' ')[0]]=m.group(1)
for r in rows: print(r)
"
