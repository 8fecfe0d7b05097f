#!/usr/bin/env python3
"""Does processing the batch in room chunks that fit the 256 MiB Infinity Cache help (X re-reads served on die)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disco_amd import _lib, synth
from disco_amd.engine import Engine
lib = _lib.load()
R, K, M, L = 1000, 4, 4, 160000
dev = torch.device('cuda', 0)
y, s_ref, n_ref = synth.make_rooms_torch(R, K, M, L, device=dev, ref_only_sn=True)
for Rc in (1000, 250, 100, 50, 25, 10):
    eng = Engine(rooms=Rc, nodes=K, mics=M, length=L, lib=lib)
    T, F = eng.T, eng.F
    mask = torch.empty((R, K, T, F), dtype=torch.float32, device=dev)
    out = torch.empty((R, K, L), dtype=torch.float32, device=dev)
    ws = torch.empty(eng.workspace_bytes(), dtype=torch.uint8, device=dev)
    def step():
        for r0 in range(0, R, Rc):
            sl = slice(r0, r0 + Rc)
            eng._chk(lib.disco_mask_oracle(eng.ctx, s_ref[sl].data_ptr(), n_ref[sl].data_ptr(), Rc * K, mask[sl].data_ptr(), None))
            eng._chk(lib.disco_tango_enhance(eng.ctx, y[sl].data_ptr(), mask[sl].data_ptr(), mask[sl].data_ptr(), out[sl].data_ptr(),
                                             None, None, ws.data_ptr(), ws.numel(), None))
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print(f'chunk={Rc}: {dt*1e3:.2f} ms/step (X chunk {Rc*20.6:.0f} MB)', flush=True)
    del eng, mask, out, ws
