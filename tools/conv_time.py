#!/usr/bin/env python3
"""Time disco_rir_convolve on a C3-sized batch (rooms x 2 sources, 16 microphones, 10 s, 4096-tap RIRs).
Usage: tools/conv_time.py [rooms] [taps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from disco_amd._engines import get_engine

rooms = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
Lh = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_sig, n_ch, L = 2 * rooms, 16, 160000
dev = 'cuda'
eng = get_engine(rooms=1, nodes=1, mics=1, length=1024)
dry = torch.randn((n_sig, L), device=dev)
rir = torch.randn((n_sig, n_ch, Lh), device=dev) * torch.exp(-6.9 * torch.arange(Lh, device=dev) / Lh)
out = torch.empty((n_sig, n_ch, L), device=dev)
p = lambda t: t.data_ptr()
def run():
    eng._chk(eng.lib.disco_rir_convolve(eng.ctx, p(dry), p(rir), n_sig, n_ch, L, Lh, p(out), L, None))
run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
# spot check against torch's FFT convolution (float64)
i, c = 1, 3
nfft = 1 << 18
ref = torch.fft.irfft(torch.fft.rfft(dry[i].double(), nfft) * torch.fft.rfft(rir[i, c].double(), nfft), nfft)[:L]
err = float((out[i, c].double() - ref).abs().max() / ref.abs().max())
print(json.dumps({'rooms': rooms, 'signals': n_sig, 'channels': n_ch, 'taps': Lh, 'ms': round(1e3 * dt, 2),
                  'out_GB': round(out.numel() * 4 / 1e9, 2), 'out_GBps': round(out.numel() * 4 / dt / 1e9, 1),
                  'audio_seconds_per_second': round(n_sig * n_ch * 10.0 / dt), 'spot_check_rel_err': err}))
