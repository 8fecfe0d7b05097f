"""Experiment: does running two half-batches of the C3 step on two HIP streams (so that one half's VALU-bound solver launches
overlap the other half's HBM-bound streaming kernels) beat one full-batch launch sequence?  Prints ms per 1000-room step."""
import sys
import time

import torch

sys.path.insert(0, '.')
from disco_amd import _lib, synth
from disco_amd.engine import Engine

cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
R, K, M, L, N, ITERS = {'C3': (1000, 4, 4, 160000, 512, 1), 'C5': (200, 8, 8, 160000, 1024, 2), 'C2': (4000, 1, 4, 160000, 512, 1)}[cfg]
print(cfg, R, K, M, N, ITERS)
dev = torch.device('cuda:0')
lib = _lib.load()
y, s, n = synth.make_rooms_torch(R, K, M, L, first_room=0, device=dev, ref_only_sn=True)


def make(r0, r1, stream):
    eng = Engine(rooms=r1 - r0, nodes=K, mics=M, length=L, n_fft=N, device=0, lib=lib)
    T, F = eng.T, N // 2 + 1
    mask = torch.empty((r1 - r0, K, T, F), dtype=torch.float32, device=dev)
    out = torch.empty((r1 - r0, K, L), dtype=torch.float32, device=dev)
    ws = torch.empty(eng.workspace_bytes(), dtype=torch.uint8, device=dev)
    st = stream.cuda_stream if stream is not None else None

    def mask_fn():
        eng._chk(lib.disco_mask_oracle(eng.ctx, s[r0:r1].data_ptr(), n[r0:r1].data_ptr(), (r1 - r0) * K, mask.data_ptr(), st))

    def enh_fn():
        if ITERS > 1:
            eng._chk(lib.disco_tango_enhance_iterated(eng.ctx, y[r0:r1].data_ptr(), mask.data_ptr(), mask.data_ptr(), ITERS,
                                                      out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), st))
            return
        eng._chk(lib.disco_tango_enhance(eng.ctx, y[r0:r1].data_ptr(), mask.data_ptr(), mask.data_ptr(), out.data_ptr(),
                                         None, None, ws.data_ptr(), ws.numel(), st))
    return eng, mask_fn, enh_fn, out


def timeit(fn, steps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


e0, m0, f0, out0 = make(0, R, None)
def full():
    m0(); f0()
print('one stream, full batch      : %.3f ms' % timeit(full))
ref = out0.clone()
del e0, m0, f0

for parts in (2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    edges = [R * i // parts for i in range(parts + 1)]
    halves = [make(edges[i], edges[i + 1], streams[i]) for i in range(parts)]

    def split_plain():
        for _, m, f, _o in halves:
            m(); f()

    def split_masks_first():
        for _, m, f, _o in halves:
            m()
        for _, m, f, _o in halves:
            f()
    print('%d streams, plain            : %.3f ms' % (parts, timeit(split_plain)))
    print('%d streams, masks first      : %.3f ms' % (parts, timeit(split_masks_first)))
    got = torch.cat([h[3] for h in halves])
    print('   max |diff| vs full batch : %.3e' % float((got - ref).abs().max()))
    del halves, streams
