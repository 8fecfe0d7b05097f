"""C5 (200 rooms x 8 x 8, 1024-pt, 2 iterations) on ONE batch under several kernel routes: ms per step, stage times, and the error of sampled
rooms against the float64 oracle (computed once, in worker processes, while the GPU runs the variants).  Test / measurement tooling.
Usage: python tools/gpu/exp_c5_variants.py out.json [rooms=200] [sample=0,100,199 | spread:N] [variants=_:_:cov_chunks:wide,... (the first two fields named round 4's room_sub / cov1_mode options, removed in round 5)  wide: -1 = disco_apply + disco_istft, 0 = one-pass filter + iSTFT, n > 0 = that with n frame pairs per run] [sample=0,100,199] [workers=N oracle processes (default: one per sampled room, up to the host's cores)]"""
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def oracle_room(args):
    import numpy as np
    yr, sr, nr, n_fft, iters = args
    from oracle import stft_oracle as so, tango_oracle as to
    s = np.zeros_like(yr); n = np.zeros_like(yr)
    s[:, 0] = sr; n[:, 0] = nr
    o = to.offline_tango_vec(yr, s, n, vads=['irm1', 'irm1'], n_fft=n_fft, hop=n_fft // 2, precision='f64', solver='eigh', extra_iters=iters - 1)
    return [so.istft(o['yf'][k], yr.shape[-1], n_fft, n_fft // 2, work_dtype=np.float64) for k in range(yr.shape[0])]


def main():
    import numpy as np
    import torch
    from disco_amd import synth
    from disco_amd.engine import Engine
    out_path = sys.argv[1]
    kv = dict(a.split('=') for a in sys.argv[2:])
    R = int(kv.get('rooms', 200))
    K, M, N, L, iters = int(kv.get('nodes', 8)), int(kv.get('mics', 8)), int(kv.get('n_fft', 1024)), 160000, int(kv.get('iters', 2))
    variants = [tuple(int(x) for x in v.split(':')) for v in kv.get('variants', '8:4:0:2,8:4:0:1,4:4:0:2').split(',')]
    sspec = kv.get('sample', '0,100,199')
    if sspec.startswith('spread:'):            # spread:N = N rooms spread evenly over the batch (first ... last)
        n_s = int(sspec.split(':')[1])
        sample = sorted({int(round(i * (R - 1) / max(n_s - 1, 1))) for i in range(n_s)})
    else:
        sample = [int(x) for x in sspec.split(',') if int(x) < R]
    steps = int(kv.get('steps', 8))
    dev = torch.device('cuda:0')
    eng = Engine(rooms=R, nodes=K, mics=M, length=L, n_fft=N, device=0)
    y, s_ref, n_ref = synth.make_rooms_torch(R, K, M, L, first_room=0, device=dev, ref_only_sn=True)
    pool = ProcessPoolExecutor(max_workers=max(1, min(len(sample), int(kv.get('workers', 0)) or (os.cpu_count() or 8) - 1)))
    futs = {r: pool.submit(oracle_room, (y[r].cpu().numpy(), s_ref[r].cpu().numpy(), n_ref[r].cpu().numpy(), N, iters)) for r in sample}
    T, F = eng.T, eng.F
    mask = torch.empty((R, K, T, F), dtype=torch.float32, device=dev)
    out = torch.empty((R, K, L), dtype=torch.float32, device=dev)
    ws = torch.empty(eng.workspace_bytes(), dtype=torch.uint8, device=dev)
    lib = eng.lib

    def step():
        eng._chk(lib.disco_mask_oracle(eng.ctx, s_ref.data_ptr(), n_ref.data_ptr(), R * K, mask.data_ptr(), None))
        if iters > 1:
            eng._chk(lib.disco_tango_enhance_iterated(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), iters, out.data_ptr(), None, None,
                                                      ws.data_ptr(), ws.numel(), None))
        else:
            eng._chk(lib.disco_tango_enhance(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), out.data_ptr(), None, None, ws.data_ptr(), ws.numel(), None))
    res = {'rooms': R, 'shape': [K, M, N], 'iters': iters, 'steps': steps, 'variants': {}}
    got = {}
    for rs, cs, ch, nf in variants:
        name = f'room_sub={rs},cov1_mode={cs},cov_chunks={ch},wide={nf}'
        eng.set_option('fuse_wide_istft', 0 if nf < 0 else 1)
        eng.set_tuning(0, ch, 0, max(nf, 0))
        if eng.workspace_bytes() > ws.numel():
            ws = torch.empty(eng.workspace_bytes(), dtype=torch.uint8, device=dev)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        ov = eng.get_option('overlap_solves')
        eng.set_option('overlap_solves', 0)
        step(); torch.cuda.synchronize()
        reps = []
        for _ in range(3):
            eng.stage_timing(True)
            step()
            reps.append(eng.stage_report())
        eng.stage_timing(False)
        eng.set_option('overlap_solves', ov)
        stages = {nm: round(sorted(r_[nm][0] for r_ in reps)[1], 3) for nm in reps[0]}
        step(); torch.cuda.synchronize()
        got[name] = {r: out[r].cpu().numpy() for r in sample}
        res['variants'][name] = {'ms_per_step': round(ms, 3), 'stages_ms': stages}
        print(name, round(ms, 3), stages, flush=True)
        json.dump(res, open(out_path, 'w'), indent=1)
    for r in sample:
        ref = futs[r].result(timeout=1500)
        for name in got:
            e = max(float(np.linalg.norm(got[name][r][k] - ref[k]) / np.linalg.norm(ref[k])) for k in range(K))
            res['variants'][name].setdefault('rel_err', {})[str(r)] = e
        json.dump(res, open(out_path, 'w'), indent=1)
    for name, v in res['variants'].items():
        errs = sorted(v['rel_err'].values())
        v['rel_err_summary'] = {'rooms': len(errs), 'worst': errs[-1], 'median': errs[len(errs) // 2],
                                'histogram_edges': [1e-6, 3e-6, 1e-5, 2e-5, 3e-5, 5e-5, 1e-4],
                                'histogram': [int(sum(1 for e in errs if lo <= e < hi)) for lo, hi in
                                              zip([0, 1e-6, 3e-6, 1e-5, 2e-5, 3e-5, 5e-5, 1e-4], [1e-6, 3e-6, 1e-5, 2e-5, 3e-5, 5e-5, 1e-4, 1e9])]}
        print(name, v['ms_per_step'], 'worst %.2e median %.2e' % (errs[-1], errs[len(errs) // 2]), v['rel_err_summary']['histogram'],
              {k_: '%.2e' % e for k_, e in v['rel_err'].items()} if len(errs) <= 12 else '', flush=True)
    json.dump(res, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
