#!/bin/bash
# Registers, spills, scratch and LDS of every kernel of a translation unit, from the compiler's own report (no GPU needed).
# Usage: tools/kernel_resources.sh [<csrc dir>] [<file.hip>] [extra flags]   (default: graphtyper_amd/csrc gtx_api.hip)
DIR=${1:-graphtyper_amd/csrc}; SRC=${2:-gtx_api.hip}; shift 2 2>/dev/null
cd "$DIR" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -c "$SRC" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  python3 -c "
import sys, re
cur = {}
for line in sys.stdin:
    m = re.search(r'remark:\s+(Function Name|VGPRs|TotalSGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)', line)
    if not m: continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = {'name': v}
    else:
        cur[k.split(' [')[0]] = v
        if k.startswith('LDS'):
            print('%-46s vgpr %3s sgpr %3s scratch %5s occ %2s sspill %4s vspill %4s lds %6s' % (re.sub(r'^_ZN(3gtx|12_GLOBAL__N_1)\d+', '', cur['name'])[:46], cur.get('VGPRs'), cur.get('TotalSGPRs'), cur.get('ScratchSize'), cur.get('Occupancy'), cur.get('SGPRs Spill'), cur.get('VGPRs Spill'), v))
"
