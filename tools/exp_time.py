#!/usr/bin/env python3
"""Time individual C-ABI stages for several library builds (exp_libs/*.so) on the GPU; one process per lib.
Usage: tools/exp_time.py stage[,stage] lib1.so lib2.so ...    (stages: stft mask istft cov1 cov2 apply1 apply2 solve1 solve2)"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
from disco_amd import _lib
from disco_amd.engine import Engine
stages = sys.argv[1].split(',')
R, K, M, L = int(os.environ.get('EXP_R', 1000)), int(os.environ.get('EXP_K', 4)), int(os.environ.get('EXP_M', 4)), 160000
lib = _lib.load()
eng = Engine(rooms=R, nodes=K, mics=M, length=L, lib=lib)
T, F = eng.T, eng.F
dev = 'cuda'
G = R * K
P2 = M + K - 1
y = torch.randn((R, K, M, L), device=dev)
sr = torch.randn((G, L), device=dev); nr = torch.randn((G, L), device=dev)
X = torch.randn((R, K, T, F, M, 2), device=dev)
z = torch.randn((R, K, T, F, 2), device=dev)
yf = torch.empty_like(z)
mask = torch.rand((R, K, T, F), device=dev)
out = torch.empty((R, K, L), device=dev)
Rss = torch.empty((R, K, F, P2, P2, 2), device=dev); Rnn = torch.empty_like(Rss)
w = torch.randn((R, K, F, P2, 2), device=dev)
p = lambda t: t.data_ptr()
calls = {
 'mask': lambda: lib.disco_mask_oracle(eng.ctx, p(sr), p(nr), G, p(mask), None),
 'stft': lambda: lib.disco_stft(eng.ctx, p(y), G, M, p(X), None),
 'cov1': lambda: lib.disco_cov_masked(eng.ctx, p(X), p(mask), None, None, 0, M, p(Rss), p(Rnn), None),
 'solve1': lambda: lib.disco_gevd_mwf_r1(eng.ctx, p(Rss), p(Rnn), G * F, M, 1.0, p(w), None, None),
 'apply1': lambda: lib.disco_apply(eng.ctx, p(X), None, p(w), M, 1, p(z), None),
 'cov2': lambda: lib.disco_cov_masked(eng.ctx, p(X), p(mask), p(z), p(z), 1, P2, p(Rss), p(Rnn), None),
 'solve2': lambda: lib.disco_gevd_mwf_r1(eng.ctx, p(Rss), p(Rnn), G * F, P2, 1.0, p(w), None, None),
 'apply2': lambda: lib.disco_apply(eng.ctx, p(X), p(z), p(w), P2, 1, p(yf), None),
 'istft': lambda: lib.disco_istft(eng.ctx, p(yf), G, p(out), None),
 'stftcov': lambda: lib.disco_stft_cov_fused(eng.ctx, p(y), p(mask), p(X), p(Rss), p(Rnn), None),
 's2cov': lambda: lib.disco_step2_cov_fused(eng.ctx, p(X), p(mask), p(w), None, p(Rss), p(Rnn), None),
 's2cov_z': lambda: lib.disco_step2_cov_fused(eng.ctx, p(X), p(mask), p(w), p(z), p(Rss), p(Rnn), None),
 's2ai': lambda: lib.disco_step2_apply_istft_fused(eng.ctx, p(X), p(w), p(w), p(out), None),
 's2apply': lambda: lib.disco_step2_apply_fused(eng.ctx, p(X), p(w), p(w), None, p(yf), None),
}
res = {}
for st in stages:
    if st.startswith('solve'):
        # realistic covariances for the solver
        calls['cov1' if st == 'solve1' else 'cov2']()
    for _ in range(2): eng._chk(calls[st]())
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); eng._chk(calls[st]()); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    res[st] = round(sorted(ts)[len(ts)//2], 3)
print('RESULT', json.dumps(res))
''' % REPO

stages = sys.argv[1]
for lib in sys.argv[2:]:
    env = dict(os.environ, DISCO_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, '-c', CHILD, stages], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT')]
    print(os.path.basename(lib), line[0][7:] if line else ('FAILED: ' + r.stderr[-400:]), flush=True)
