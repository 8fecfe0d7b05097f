#!/usr/bin/env python3
"""Time the on-GPU evaluation metrics on a C3-sized batch of time signals.  Usage: tools/metrics_time.py [n_sig] [L]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from disco_amd import metrics as gm

n_sig = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 160000
fs = 16000
dev = 'cuda'
s = torch.randn((n_sig, L), device=dev)
n = torch.randn((n_sig, L), device=dev) * 0.5
s[:, :fs] = 0
res = {}
for name, fn in (('snr', lambda: gm.snr(s, n, start=fs)), ('si_sdr', lambda: gm.si_sdr(s, n, start=fs)),
                 ('fw_snr', lambda: gm.fw_snr(s, n, fs, start=fs))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[name] = {'ms': round(1e3 * dt, 2), 'GB_read': round(2 * n_sig * (L - fs) * 4 / 1e9, 2),
                 'GBps': round(2 * n_sig * (L - fs) * 4 / dt / 1e9, 1)}
print(json.dumps({'n_sig': n_sig, 'L': L, **res}))
