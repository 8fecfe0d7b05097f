#!/usr/bin/env python3
"""Does splitting the batch into S sub-batches on S HIP streams help (tails + latency-bound solver overlap)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from disco_amd import _lib, synth
from disco_amd.engine import Engine
lib = _lib.load()
R, K, M, L = 1000, 4, 4, 160000
dev = torch.device('cuda', 0)
y, s_ref, n_ref = synth.make_rooms_torch(R, K, M, L, device=dev, ref_only_sn=True)
for S in (1, 2, 4):
    Rs = R // S
    engs = [Engine(rooms=Rs, nodes=K, mics=M, length=L, lib=lib) for _ in range(S)]
    T, F = engs[0].T, engs[0].F
    streams = [torch.cuda.Stream() for _ in range(S)]
    masks = [torch.empty((Rs, K, T, F), dtype=torch.float32, device=dev) for _ in range(S)]
    outs = [torch.empty((Rs, K, L), dtype=torch.float32, device=dev) for _ in range(S)]
    wss = [torch.empty(engs[0].workspace_bytes(), dtype=torch.uint8, device=dev) for _ in range(S)]
    def step():
        for i in range(S):
            st = streams[i].cuda_stream
            e = engs[i]
            sl = slice(i * Rs, (i + 1) * Rs)
            e._chk(lib.disco_mask_oracle(e.ctx, s_ref[sl].data_ptr(), n_ref[sl].data_ptr(), Rs * K, masks[i].data_ptr(), st))
            e._chk(lib.disco_tango_enhance(e.ctx, y[sl].data_ptr(), masks[i].data_ptr(), masks[i].data_ptr(), outs[i].data_ptr(),
                                           None, None, wss[i].data_ptr(), wss[i].numel(), st))
    for _ in range(2): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f'S={S}: {dt*1e3:.2f} ms/step  {R*K*T/dt/1e6:.1f} M node-frames/s', flush=True)
    del engs, masks, outs, wss
