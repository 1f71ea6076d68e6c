#!/bin/bash
# usage: scripts/kernel_resources.sh [name-pattern]: registers / scratch / LDS / occupancy of the device kernels (gfx950), from
# the compiler's own report (-Rpass-analysis=kernel-resource-usage); nothing is installed, the object goes to /tmp
cd "$(dirname "$0")/../smoothmesh_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -c smgpu.hip -o /tmp/smgpu_res.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ''
cur = None
rows = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z][\w \[\]/]+?): (\d+) \[-Rpass', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
import subprocess
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip().split('(')[0]
    if pat in name:
        print(f\"{name[:70]:70s} VGPR {v.get('VGPRs', -1):4d} AGPR {v.get('AGPRs', -1):3d} SGPR {v.get('TotalSGPRs', -1):4d} spillV {v.get('VGPRs Spill', -1):3d} spillS {v.get('SGPRs Spill', -1):3d} scratch {v.get('ScratchSize [bytes/lane]', -1):5d} occ {v.get('Occupancy [waves/SIMD]', -1):2d} LDS {v.get('LDS Size [bytes/block]', -1)}\")
" "$1"
