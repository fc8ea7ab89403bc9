#!/bin/bash
# Register / spill / occupancy table of every kernel of libvtmhip.so as the compiler reports it (no GPU needed).
# usage: scripts/resource_usage.sh > profiles/<tag>_kernel_resources.txt
cd "$(dirname "$0")/.."
for f in vtm_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -ffp-contract=off --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 | python3 -c "
import sys, re, subprocess
cur = None; d = {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m: cur = m.group(1); d = {}; continue
    m = re.search(r'remark: +([A-Za-z ]+?)( \[[^\]]*\])?: (\d+)', l)
    if m and cur: d[m.group(1)] = m.group(3)
    if 'LDS Size' in l and cur:
        name = subprocess.run(['c++filt', cur], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(anonymous namespace\)::', '', name); name = re.sub(r'\(.*', '', name); name = re.sub(r'^void ', '', name)
        print('%-48s VGPRs %3s  spilled %3s  scratch %4s B  waves/SIMD %s  SGPRs %3s  static LDS %6s B' % (name[:48], d.get('VGPRs'), d.get('VGPRs Spill'), d.get('ScratchSize'), d.get('Occupancy'), d.get('TotalSGPRs'), d.get('LDS Size')))
        cur = None
"
done
