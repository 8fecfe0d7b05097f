#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage.
Usage: tools/kernel_resources.py [substring ...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, 'disco_amd', 'csrc', 'disco_hip.hip')
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', '-c', '-fPIC', '-Rpass-analysis=kernel-resource-usage',
       '-o', '/tmp/_res.o', src] + [a for a in sys.argv[1:] if a.startswith('-D')]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r'remark: (?:.*?:\d+:\d+: )?\s*(Function Name|Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\S+)', line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k in ('Function Name', 'Name'):
        cur = {'name': v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(' ')[0]] = v
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.splitlines()
filt = [a for a in sys.argv[1:] if not a.startswith('-D')]
print(f"{'kernel':60s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'occ':>4s} {'LDS':>7s}")
for r, n in zip(rows, names):
    n = re.sub(r'HIP_vector_type<float, 2u>', 'c32', n)
    n = re.sub(r'\(.*$', '', n).replace('void disco::', '')
    if filt and not any(f in n for f in filt):
        continue
    print(f"{n[:60]:60s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('SGPRs','?'):>5s} {r.get('ScratchSize','?'):>7s} {r.get('Occupancy','?'):>4s} {r.get('LDS','?'):>7s}")
