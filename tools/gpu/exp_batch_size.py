"""C3-shaped rooms (4 nodes x 4 mics, 512-pt, L = 160000) in batches of different sizes: microseconds per room and per stage.  Does a batch
whose spectra fit the 256 MB memory-side cache re-read them faster than HBM allows?  Test / measurement tooling.
Usage: python tools/gpu/exp_batch_size.py out.json [rooms=4,8,16,32,64,125,250,1000]"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def main():
    import torch
    from disco_amd import synth
    from disco_amd.engine import Engine
    out_path = sys.argv[1]
    kv = dict(a.split('=') for a in sys.argv[2:])
    sizes = [int(x) for x in kv.get('rooms', '4,8,16,32,64,125,250,1000').split(',')]
    K, M, N, L = 4, 4, 512, 160000
    dev = torch.device('cuda:0')
    res = {}
    for R in sizes:
        eng = Engine(rooms=R, nodes=K, mics=M, length=L, n_fft=N, device=0)
        y, s_ref, n_ref = synth.make_rooms_torch(R, K, M, L, first_room=0, device=dev, ref_only_sn=True)
        mask = torch.empty((R, K, eng.T, eng.F), dtype=torch.float32, device=dev)
        out = torch.empty((R, K, L), dtype=torch.float32, device=dev)
        ws = torch.empty(eng.workspace_bytes(), dtype=torch.uint8, device=dev)
        lib = eng.lib
        eng._chk(lib.disco_mask_oracle(eng.ctx, s_ref.data_ptr(), n_ref.data_ptr(), R * K, mask.data_ptr(), None))

        def step():
            eng._chk(lib.disco_tango_enhance(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None))
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        steps = max(10, min(200, 4000 // R))
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        us_room = 1e6 * (time.perf_counter() - t0) / steps / R
        eng.stage_timing(True)
        for _ in range(5):
            step()
        rep = eng.stage_report()
        eng.stage_timing(False)
        stages = {nm: round(1e3 * v[0] / max(v[2], 1), 3) for nm, v in rep.items()}      # us per room
        res[R] = {'us_per_room': round(us_room, 2), 'stage_us_per_room': stages}
        print(R, res[R], flush=True)
        json.dump(res, open(out_path, 'w'), indent=1)
        del eng, y, s_ref, n_ref, mask, out, ws
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
