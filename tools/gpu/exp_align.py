"""Is the bimodal time of C3's step-2 kernels (k_step2_apply_istft 4.39 / 4.86 ms, k_step2_cov_fused 4.42 / 4.68 ms between processes on ONE box)
a matter of where the buffers lie?  One process, one batch; the workspace (X inside it), the output and the mask are placed at different byte
offsets inside over-sized allocations and the stage times are read per placement.  Usage: python tools/gpu/exp_align.py [rooms=1000]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disco_amd import synth
from disco_amd.engine import Engine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K, M, L = 4, 4, 160000
dev = torch.device('cuda:0')
eng = Engine(rooms=R, nodes=K, mics=M, length=L, device=0)
lib = eng.lib
y, s_ref, n_ref = synth.make_rooms_torch(R, K, M, L, device=dev, ref_only_sn=True)
T, F = eng.T, eng.F
need = eng.workspace_bytes()
SL = 1 << 26
ws_big = torch.empty(need + SL, dtype=torch.uint8, device=dev)
out_big = torch.empty(R * K * L + SL // 4, dtype=torch.float32, device=dev)
mask_big = torch.empty(R * K * T * F + SL // 4, dtype=torch.float32, device=dev)
eng.set_option('overlap_solves', 0)
print('base addresses mod 2^21: ws %x out %x mask %x y %x' % tuple(t.data_ptr() & ((1 << 21) - 1) for t in (ws_big, out_big, mask_big, y)), flush=True)


def run(pw, po, pm, reps=3):
    ws = ws_big[pw:pw + need]
    out = out_big[po // 4:po // 4 + R * K * L]
    mask = mask_big[pm // 4:pm // 4 + R * K * T * F]
    def step():
        eng._chk(lib.disco_mask_oracle(eng.ctx, s_ref.data_ptr(), n_ref.data_ptr(), R * K, mask.data_ptr(), None))
        eng._chk(lib.disco_tango_enhance(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None))
    step(); torch.cuda.synchronize()
    acc = {}
    for _ in range(reps):
        eng.stage_timing(True)
        step()
        rep = eng.stage_report()
        for k_, v in rep.items():
            acc.setdefault(k_, []).append(v[0])
    eng.stage_timing(False)
    return {k_: round(sorted(v)[len(v) // 2], 3) for k_, v in acc.items()}


for pw, po, pm in [(0, 0, 0), (4096, 0, 0), (65536, 0, 0), (1 << 20, 0, 0), (1 << 21, 0, 0), (3 << 20, 0, 0), (0, 4096, 0), (0, 65536, 0), (0, 1 << 20, 0), (0, 1 << 21, 0),
                   (0, 0, 4096), (0, 0, 1 << 20), (1 << 20, 1 << 20, 1 << 20), (256, 256, 256), (8192, 16384, 32768), (0, 0, 0)]:
    r = run(pw, po, pm)
    print('ws+%-8d out+%-8d mask+%-8d' % (pw, po, pm), r, 'sum %.3f' % sum(r.values()), flush=True)
