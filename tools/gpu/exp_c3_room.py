"""C3 (4 nodes x 4 mics, 512-pt): the rooms the whole-batch sweep found farthest from the float64 oracle (profiles/r05_o_parity_C3_all_1000.json), taken
apart: output error, error of the step-1 output z, and the same under other launch geometries (frames per STFT wave / partial-sum chunk counts: the
lengths of the float32 sums).  A small batch (first_room ... first_room + R) pinned to the large batch's geometry.  Test / measurement tooling.
Usage: python tools/gpu/exp_c3_room.py out.json first_room=436 rooms=8 watch=439"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)


def oracle_room(args):
    import numpy as np
    yr, sr, nr, n_fft = args
    from oracle import stft_oracle as so, tango_oracle as to
    s = np.zeros_like(yr); n = np.zeros_like(yr)
    s[:, 0] = sr; n[:, 0] = nr
    o = to.offline_tango_vec(yr, s, n, vads=['irm1', 'irm1'], n_fft=n_fft, hop=n_fft // 2, precision='f64', solver='eigh')
    K = yr.shape[0]
    out = [so.istft(o['yf'][k], yr.shape[-1], n_fft, n_fft // 2, work_dtype=np.float64) for k in range(K)]
    z = [np.asarray(o['z_y'][k]) for k in range(K)]                       # (F, T)
    # conditioning of the pencils: cond(Rnn) of step 2 per (node, bin)
    cond = [float(np.max(np.linalg.cond(np.asarray(o['Rnn_glo'][k])))) for k in range(K)]
    return out, z, cond


def main():
    import numpy as np
    import torch
    from disco_amd import synth
    from disco_amd.engine import Engine
    out_path = sys.argv[1]
    kv = dict(a.split('=') for a in sys.argv[2:])
    first, R, watch = int(kv.get('first_room', 436)), int(kv.get('rooms', 8)), [int(x) for x in kv.get('watch', '439').split(',')]
    K, M, N, L = 4, 4, 512, 160000
    dev = torch.device('cuda:0')
    eng = Engine(rooms=R, nodes=K, mics=M, length=L, n_fft=N, device=0)
    y, s_ref, n_ref = synth.make_rooms_torch(R, K, M, L, first_room=first, device=dev, ref_only_sn=True)
    pool = ProcessPoolExecutor(max_workers=len(watch))
    futs = {r: pool.submit(oracle_room, (y[r - first].cpu().numpy(), s_ref[r - first].cpu().numpy(), n_ref[r - first].cpu().numpy(), N)) for r in watch}
    T, F = eng.T, eng.F
    mask = torch.empty((R, K, T, F), dtype=torch.float32, device=dev)
    out = torch.empty((R, K, L), dtype=torch.float32, device=dev)
    z = torch.empty((R, K, T, F, 2), dtype=torch.float32, device=dev)
    ws = torch.empty(eng.workspace_bytes(), dtype=torch.uint8, device=dev)
    lib = eng.lib
    res = {'first_room': first, 'rooms': R, 'variants': {}}
    got = {}
    # (stft frames per wave, cov chunks, step-2 chunks): 0 = the heuristic of THIS small batch; 80 / 2 / .. = what 1000 rooms get
    for name, tun in (('heuristic of 8 rooms', (0, 0, 0)), ('frames per wave 80 (the 1000-room geometry)', (80, 0, 0)), ('frames per wave 40', (40, 0, 0)),
                      ('frames per wave 20', (20, 0, 0)), ('frames per wave 80, step-2 chunks 4', (80, 0, 4)), ('frames per wave 80, step-2 chunks 8', (80, 0, 8)),
                      ('frames per wave 20, step-2 chunks 8', (20, 0, 8))):
        eng.set_tuning(tun[0], tun[1], tun[2], 0)
        if eng.workspace_bytes() > ws.numel():
            ws = torch.empty(eng.workspace_bytes(), dtype=torch.uint8, device=dev)
        eng._chk(lib.disco_mask_oracle(eng.ctx, s_ref.data_ptr(), n_ref.data_ptr(), R * K, mask.data_ptr(), None))
        eng._chk(lib.disco_tango_enhance(eng.ctx, y.data_ptr(), mask.data_ptr(), mask.data_ptr(), out.data_ptr(), z.data_ptr(), None, ws.data_ptr(), ws.numel(), None))
        torch.cuda.synchronize()
        got[name] = ({r: out[r - first].cpu().numpy() for r in watch}, {r: torch.view_as_complex(z[r - first]).cpu().numpy() for r in watch})
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    for r in watch:
        ref_out, ref_z, cond = futs[r].result(timeout=900)
        for name, (o_, z_) in got.items():
            res['variants'].setdefault(name, {})[str(r)] = {
                'out_rel_per_node': [rel(o_[r][k], ref_out[k]) for k in range(K)],
                'z_rel_per_node': [rel(z_[r][k].T, ref_z[k]) for k in range(K)]}            # GPU z is (T, F), the oracle's (F, T)
        res.setdefault('cond_Rnn_step2_max_per_node', {})[str(r)] = cond
    for name, v in res['variants'].items():
        print(name, {r: ('out %.2e' % max(x['out_rel_per_node']), 'z %.2e' % max(x['z_rel_per_node'])) for r, x in v.items()}, flush=True)
    print('cond', res['cond_Rnn_step2_max_per_node'])
    json.dump(res, open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
