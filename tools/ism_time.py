#!/usr/bin/env python3
"""Time disco_ism_rir for a C3-sized data set (rooms x 2 sources x 16 microphones, max_order 20, 4096 taps) and chain it into
disco_rir_convolve.  Usage: tools/ism_time.py [rooms] [max_order]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from disco_amd._engines import get_engine

rooms = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
order = int(sys.argv[2]) if len(sys.argv) > 2 else 20
S, Q, Lh, L = 2, 16, 4096, 160000
dev = 'cuda'
torch.manual_seed(0)
eng = get_engine(rooms=1, nodes=1, mics=1, length=1024)
dims = torch.stack([3 + 5 * torch.rand(rooms), 3 + 2 * torch.rand(rooms), 2.5 + 0.5 * torch.rand(rooms)], 1).to(dev)
absorb = (0.2 + 0.4 * torch.rand(rooms)).to(dev)
src = ((0.2 + 0.6 * torch.rand(rooms, S, 3)).to(dev) * dims[:, None]).contiguous()
mic = ((0.2 + 0.6 * torch.rand(rooms, Q, 3)).to(dev) * dims[:, None]).contiguous()
rir = torch.empty((rooms, S, Q, Lh), device=dev)
p = lambda t: t.data_ptr()
def run():
    eng._chk(eng.lib.disco_ism_rir(eng.ctx, p(dims), p(absorb), p(src), p(mic), rooms, S, Q, order, 16000.0, 343.0, p(rir), Lh, None))
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
dry = torch.randn((rooms * S, L), device=dev)
out = torch.empty((rooms * S, Q, L), device=dev)
eng._chk(eng.lib.disco_rir_convolve(eng.ctx, p(dry), p(rir), rooms * S, Q, L, Lh, p(out), L, None))
torch.cuda.synchronize()
n_img = sum(1 for a in range(-order, order + 1) for b in range(-order, order + 1) for c in range(-order, order + 1) if abs(a) + abs(b) + abs(c) <= order)
print(json.dumps({'rooms': rooms, 'rirs': rooms * S * Q, 'max_order': order, 'images_per_rir': n_img, 'taps': Lh, 'ism_ms': round(1e3 * dt, 1),
                  'images_per_s': round(rooms * S * Q * n_img / dt), 'energy_first_rir': float((rir[0, 0, 0] ** 2).sum()),
                  'reverberated_finite': bool(torch.isfinite(out).all())}))
