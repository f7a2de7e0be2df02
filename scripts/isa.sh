#!/bin/bash
# compile lep_gpu.hip for gfx950, print per-kernel resource usage and dump named kernels' ISA to /tmp/st/<name>.s
mkdir -p /tmp/st; cd /root/repo/lepton_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c lep_gpu.hip -o /tmp/st/lep_gpu.o -save-temps=obj -Rpass-analysis=kernel-resource-usage $EXTRA 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r'Function Name: (\S+)',line)
    if m:
        n=m.group(1); k=re.search(r'(lep_\w+?kernel(ILb\d)?)',n); cur=k.group(1) if k else n[:40]; print(); print(cur,end=': ')
    for key in ('VGPRs:','TotalSGPRs:','ScratchSize','Occupancy','SGPRs Spill','VGPRs Spill','LDS Size'):
        m=re.search(re.escape(key)+r'[^:]*:? *(\d+)',line)
        if m and key in line: print(key.split()[0].rstrip(':')+'='+m.group(1),end=' ')
print()
"
S=/tmp/st/lep_gpu-hip-amdgcn-amd-amdhsa-gfx950.s
for k in "$@"; do
  awk -v k="$k" '$0 ~ "^_ZN.*" k ".*:" {f=1} f{print} /s_endpgm/{if(f) exit}' $S > /tmp/st/$k.s
  echo "$k: lines $(wc -l < /tmp/st/$k.s) valu $(grep -c '^\s*v_' /tmp/st/$k.s) salu $(grep -c '^\s*s_' /tmp/st/$k.s) gload $(grep -c global_load /tmp/st/$k.s) gstore $(grep -c global_store /tmp/st/$k.s) ds $(grep -c '^\s*ds_' /tmp/st/$k.s) scratch $(grep -c scratch_ /tmp/st/$k.s) branches $(grep -c s_cbranch /tmp/st/$k.s) waitcnt $(grep -c s_waitcnt /tmp/st/$k.s)"
done
