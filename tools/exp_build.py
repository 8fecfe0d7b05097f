#!/usr/bin/env python3
"""Build experimental variants of the library into gpurun_out/exp/ (A/B of kernel ablations / tunables).
Usage: tools/exp_build.py name1:-DFOO=1,-DBAR=2 name2:...   ->  exp_libs/libdisco_<name>.so"""
import os
import subprocess
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(REPO, 'exp_libs')
os.makedirs(out_dir, exist_ok=True)
procs = []
for spec in sys.argv[1:]:
    name, _, flags = spec.partition(':')
    flags = [f for f in flags.split(',') if f]
    out = os.path.join(out_dir, f'libdisco_{name}.so')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-slp-vectorize', '-shared', '-fPIC', '-o', out,
           os.path.join(REPO, 'disco_amd', 'csrc', 'disco_hip.hip')] + flags
    procs.append((name, subprocess.Popen(cmd)))
for name, p in procs:
    rc = p.wait()
    print(name, 'rc', rc)
