#!/usr/bin/env python3
"""Does a sub-batch whose STFT fits the 256 MiB Infinity Cache beat one big launch per stage?

The C3 step writes X (20.6 MB per room) in stft_cov and reads it back twice (step-2 covariance, filter + iSTFT).  With all
1000 rooms in one launch per stage every read of X comes from HBM.  Here the same rooms go through the whole path in
sub-batches of `rs` rooms (one context of `rs` rooms, called R / rs times on slices of the same arrays), so that a
sub-batch's X (rs x 20.6 MB) may still sit in the memory-side cache when the next stage asks for it.
Usage: exp_subbatch.py [R] [rs ...]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from disco_amd import _lib, synth  # noqa: E402
from disco_amd.engine import Engine  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 960
sizes = [int(a) for a in sys.argv[2:]] or [R, 4, 6, 8, 12, 16, 32, 96]
K, M, L = 4, 4, 160000
dev = torch.device('cuda', 0)
lib = _lib.load()
y, s_ref, n_ref = synth.make_rooms_torch(R, K, M, L, first_room=0, device=dev, ref_only_sn=True)
out_ref = None
for rs in sizes:
    if R % rs:
        continue
    eng = Engine(rooms=rs, nodes=K, mics=M, length=L, device=0, lib=lib)
    T, F = eng.T, eng.F
    mask = torch.empty((R, K, T, F), dtype=torch.float32, device=dev)
    out = torch.empty((R, K, L), dtype=torch.float32, device=dev)
    ws = torch.empty(eng.workspace_bytes(), dtype=torch.uint8, device=dev)
    G = rs * K

    def step():
        for r0 in range(0, R, rs):
            eng._chk(lib.disco_mask_oracle(eng.ctx, s_ref[r0].data_ptr(), n_ref[r0].data_ptr(), G, mask[r0].data_ptr(), None))
            eng._chk(lib.disco_tango_enhance(eng.ctx, y[r0].data_ptr(), mask[r0].data_ptr(), mask[r0].data_ptr(), out[r0].data_ptr(),
                                             None, None, ws.data_ptr(), ws.numel(), None))

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        step()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    if out_ref is None:
        out_ref = out.clone()
        err = 0.0
    else:
        err = float((out - out_ref).norm() / out_ref.norm())
    print(f'rooms per sub-batch {rs:5d} ({rs * 20.6:7.1f} MB of X): {dt * 1e3:8.3f} ms per {R}-room step '
          f'(host enqueue {t_host * 1e3:7.3f} ms), {R * K * T / dt / 1e6:7.1f} M node-frames/s, rel diff vs first {err:.1e}', flush=True)
    del eng, mask, out, ws
