#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage.
Usage: tools/kernel_resources.py [-DFOO=1 ...] [substring ...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from disco_amd import build as _b

# the remarks hipcc printed for every translation unit (disco_amd/build.py keeps them next to the objects);
# -D switches go through DISCO_CXXFLAGS, as for the build itself
extra = [a for a in sys.argv[1:] if a.startswith('-D')]
if extra:
    os.environ['DISCO_CXXFLAGS'] = ' '.join(extra)
_b.build_hip(verbose=False)
import hashlib
tag = hashlib.sha1(' '.join(extra).encode()).hexdigest()[:8] if extra else 'default'
out = ''.join(open(os.path.join(_b.OBJ, f)).read() for f in sorted(os.listdir(_b.OBJ)) if f.endswith('.' + tag + '.remarks'))
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
