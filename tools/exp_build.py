#!/usr/bin/env python3
"""Build experimental variants of the library (A/B of kernel ablations / tunables).
Usage: tools/exp_build.py name1:-DFOO=1,-DBAR=2 name2:...   ->  exp_libs/libdisco_<name>.so
Each variant goes through disco_amd/build.py with DISCO_CXXFLAGS (objects are cached per flag set)."""
import os
import shutil
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from disco_amd import build as b

out_dir = os.path.join(REPO, 'exp_libs')
os.makedirs(out_dir, exist_ok=True)
keep = b.OUT + '.keep'
if os.path.exists(b.OUT):
    shutil.copy2(b.OUT, keep)
try:
    for spec in sys.argv[1:]:
        name, _, flags = spec.partition(':')
        os.environ['DISCO_CXXFLAGS'] = ' '.join(f for f in flags.split(',') if f)
        b.build_hip(verbose=False)
        shutil.copy2(b.OUT, os.path.join(out_dir, f'libdisco_{name}.so'))
        print(name, 'built')
finally:
    os.environ.pop('DISCO_CXXFLAGS', None)
    if os.path.exists(keep):
        shutil.move(keep, b.OUT)
